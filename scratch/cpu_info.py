import os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
    try:
        print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), flush=True)
    except Exception as e:
        print("no cgroup cpu.max", e, flush=True)
    import bench
    t = time.perf_counter()
    from oracle import oracle as O
    O.build()
    import multiprocessing as mp
    import numpy as np
    cores = min(len(os.sched_getaffinity(0)), int(sys.argv[1]) if len(sys.argv) > 1 else 10**6)
    with mp.get_context("spawn").Pool(cores) as pool:
        print("pool of", cores, "up after", time.perf_counter() - t, flush=True)
        for Ns in (512, 1024, 1536):
            t1 = time.perf_counter()
            r = pool.map(bench._cpu_eval_worker, [(3, 1, Ns, 8, 5, i) for i in range(cores)])
            print(Ns, "concurrent", cores, "mean %.2f s wall %.2f s" % (float(np.mean(r)), time.perf_counter() - t1), flush=True)


if __name__ == "__main__":
    main()
