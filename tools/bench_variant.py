"""bench.py against an experiment build of the library (tools/build_variant.sh NAME ... -> csrc/exp/libsavit_NAME.so):
SAVIT_EXP_LIB=NAME python tools/bench_variant.py [bench.py flags].  Dev tool for same-box A/B runs."""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import savit_amd  # noqa: F401,E402
from savit_amd import lib as _l  # noqa: E402
if os.environ.get("SAVIT_EXP_LIB"):
    _l.LIB_PATH = os.path.join(os.path.dirname(_l.LIB_PATH), "exp", "libsavit_%s.so" % os.environ["SAVIT_EXP_LIB"])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
