"""region F of bench.py alone (configs[3] trained through the CLI, one process): prints the CLI's search-stats lines.
   python scratch/r04_regionF_only.py [restarts]"""
import os, sys, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["BENCH_KEEP_STDERR"] = "1"
import bench
with tempfile.TemporaryDirectory() as t:
    print(json.dumps(bench.pca8_train_ranks_cli(0, 0, 1, 0, t, restarts=int(sys.argv[1]) if len(sys.argv) > 1 else 4)))
