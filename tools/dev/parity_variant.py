#!/usr/bin/env python3
"""Developer helper (GPU box): the row-walk and soak parity tests against a library VARIANT (tools/bin/<name>.so).
  VFGS_ALLOW_DEV_BUILD=1 python3 tools/dev/parity_variant.py <name> [pytest args]"""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
os.environ["VFGS_ALLOW_DEV_BUILD"] = "1"
import torch  # noqa: F401  (one HIP runtime: torch first)
from versatilefilmgrain_amd import hw
hw.load(ROOT / "tools" / "bin" / (sys.argv[1] + ".so"))
import pytest
sys.exit(pytest.main(["-x", "-q", "-m", "gpu", str(ROOT / "tests" / "test_gpu_rowwalk.py"), str(ROOT / "tests" / "test_gpu_parity.py")] + sys.argv[2:]))
