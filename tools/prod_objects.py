"""Object files of the PRODUCTION library, from build.SOURCES (never `ls _obj/*.o`: an object of a source that left the
build - a pit_latent.o of round 4 - must not be linked into a diagnostic library), minus the bases given as arguments.
Refuses objects whose recorded flags are not the production flags.   python tools/prod_objects.py pit_block"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from position_induced_transformer_amd import build  # noqa: E402

skip = set(sys.argv[1:])
out = []
for src in build.SOURCES:
    base = src.replace(".hip", "")
    if base in skip:
        continue
    obj = os.path.join(build.CSRC, "_obj", base + ".o")
    try:
        with open(obj + ".flags") as f:
            flags = f.read()
    except OSError:
        sys.exit(f"{obj}: no production object (run python -m position_induced_transformer_amd.build first)")
    if flags != " ".join(build.FLAGS):
        sys.exit(f"{obj}: built with other flags ({flags!r}); rebuild the production library first")
    out.append(obj)
print(" ".join(out))
