#!/bin/bash
# interactive_mode's reader / device / writer pipeline (csrc/host/interactive_io.c) under ThreadSanitizer and
# AddressSanitizer+UBSan on the CPU (the device stage is the stand-in of tests/c/io_loop_driver.c): 60 000 points, text and
# binary framing, 0 / 2 / 4 helper threads.  usage: bash tools/sanitize_io_loop.sh   (needs gcc only; writes under build/)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/sanitize
mkdir -p "$out"
for san in thread address,undefined; do
  exe=$out/io_${san%%,*}
  gcc -std=gnu99 -O1 -g -fsanitize=$san -I "$root/include" -I "$root/madaiemulator_amd/csrc/host" -o "$exe" \
      "$root/tests/c/io_loop_driver.c" "$root/madaiemulator_amd/csrc/host/interactive_io.c" -lm -lpthread
  python3 - "$exe" <<'PY'
import numpy as np, subprocess, os, sys
exe = sys.argv[1]
rng = np.random.default_rng(5)
X = rng.normal(size=(60000, 3)) * 10.0 ** rng.integers(-3, 4, size=(60000, 1))
text = ("\n".join(" ".join(repr(float(v)) for v in row) for row in X) + "\n").encode()
for threads in ("0", "2", "4"):
    for binary in ("0", "1"):
        p = subprocess.run([exe, "3", "2", "2", binary], input=text if binary == "0" else X.tobytes(), capture_output=True,
                           timeout=600, env=dict(os.environ, GPEMU_IO_THREADS=threads))
        ok = p.returncode == 0 and b"Sanitizer" not in p.stderr and len(p.stdout) == (5097891 if binary == "0" else 1920000)
        print(os.path.basename(exe), "threads", threads, "binary", binary, "ok" if ok else "FAILED: " + p.stderr.decode()[-400:])
        if not ok:
            sys.exit(1)
PY
done
