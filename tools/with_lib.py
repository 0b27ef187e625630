"""python tools/with_lib.py <library.so> <script.py> [args]: run a tool script against an explicitly named build of the HIP library
(A/B timing of alternative builds).  The product loader (anatomask_amd/hip.py) honours no environment variable."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anatomask_amd import hip  # noqa: E402

hip.use_library(os.path.abspath(sys.argv[1]))
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
