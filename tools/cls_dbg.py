import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from suggest_amd import NGramIndex, IndexDescription, synth, _lib
for build in ("host", "device"):
    blob, offs = synth.make_dict(60000, seed=1)
    ix = NGramIndex(blob=blob, offs=offs, description=IndexDescription(**synth.DESCRIPTION), build=build)
    out = (ctypes.c_uint64 * 8)()
    rc = _lib.lib().sg_debug_class_store(ix._h, out)
    print(build, rc, list(out), flush=True)
