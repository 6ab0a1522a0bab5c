"""The plain-C oracle (oracle/hifigan_oracle.c) against the reference's goldens and the python
oracle.  CPU only; the C file is built by __graft_entry__.build() / `make -C oracle`."""
import subprocess

import numpy as np
import pytest

from conftest import REPO, oracle_config
from oracle import c_oracle
from oracle import hifigan_oracle as orc

from iris._weights import weight_blob


@pytest.fixture(scope="module", autouse=True)
def _built():
    subprocess.run(["make", "-C", str(REPO / "oracle"), "all"], check=True, capture_output=True)


@pytest.mark.parametrize("case", ["v1_default_T4_taps", "small_cfg_B3_T19"])
def test_c_oracle_matches_reference_goldens(case, golden, case_setup):
    cfg, sd = case_setup(case)
    g = golden(case)
    # the C oracle takes already-folded weights in the C-ABI's blob order
    got = c_oracle.generator_forward_c(cfg, weight_blob(cfg, sd), g["mel"])
    assert got.shape == g["wav"][:, 0, :].shape
    assert np.abs(got - g["wav"][:, 0, :]).max() <= 2e-6
