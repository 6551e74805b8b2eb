"""Whole-step A/B of a tuning switch on ONE box: python tools/step_ab.py <scratch liblvae_hip.so> [bench.py arguments]
Loads the given library (a -DLVAE_TUNING_ENV build from tools/step_ab.sh, whose switches are read from the environment) instead of the
product one and runs bench.py's main(). Measurement tooling only."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
lib = os.path.abspath(sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[2:]
import bench  # noqa: E402  (imports the package)
from lvae_amd import _C  # noqa: E402
_C.LIB_PATH = lib
bench.main()
