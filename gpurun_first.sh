set -x
cd $GRAFT_REPO_ROOT
rocminfo | grep -E "gfx|Marketing" | head -4
nproc; free -g | head -2
python -c "
import ctypes, sys
sys.path.insert(0,'.')
from mmwave_msc_amd import _lib
L=_lib.load(); print(L.mmw_version())
"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -30
