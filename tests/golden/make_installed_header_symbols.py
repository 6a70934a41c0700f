"""names of the functions the reference's INSTALLED headers declare (CMakeLists.txt:99: libEmu/{emulate-fns,emulator,
estimate_threaded,estimator-fns,regression}.h; src/CMakeLists.txt:50-51: modelstruct.h optstruct.h emulator_struct.h
multi_modelstruct.h multivar_support.h; plus libEmu/maxmultimin.h and resultstruct.h, which those include) ->
installed_header_symbols.txt, one "header name" pair per line.  Run in the build container (reads /root/reference)."""
import os, re
REF = "/root/reference/src"
HEADERS = ["libEmu/emulate-fns.h", "libEmu/emulator.h", "libEmu/estimate_threaded.h", "libEmu/estimator-fns.h", "libEmu/regression.h",
           "libEmu/maxmultimin.h", "modelstruct.h", "optstruct.h", "emulator_struct.h", "multi_modelstruct.h", "multivar_support.h",
           "resultstruct.h"]
out = []
for h in HEADERS:
    text = open(os.path.join(REF, h)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    for m in re.finditer(r"^[A-Za-z_][A-Za-z0-9_ \t\*]*?[ \t\*]([A-Za-z_][A-Za-z0-9_]*)[ \t]*\(", text, flags=re.M):
        name = m.group(1)
        if name not in ("if", "while", "for", "switch", "return", "sizeof"):
            out.append((h, name))
seen = set()
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "installed_header_symbols.txt"), "w") as f:
    for h, n in out:
        if n not in seen:
            seen.add(n)
            f.write(f"{h} {n}\n")
print(len(seen), "names")
