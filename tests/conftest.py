import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Parameter sets used across the suite.
FAST = {"n": 2, "nu_1": 6, "nu_2": 2, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
        "t_exp_right": 8, "instances": 1, "db_item_size": 8192}          # util.rs:122-137
FAST56 = dict(FAST, t_exp_right=56, nu_2=3)                              # exercises the 1-bit gadget + stop_round
C1 = {"n": 2, "nu_1": 9, "nu_2": 5, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
      "t_exp_right": 56, "instances": 1, "db_item_size": 256}
P2 = {"n": 2, "nu_1": 9, "nu_2": 6, "p": 256, "q2_bits": 20, "t_gsw": 8, "t_conv": 4, "t_exp_left": 8,
      "t_exp_right": 56, "instances": 1, "db_item_size": 8192}           # CFG_20_256, util.rs:7-20
C2 = dict(C1, nu_2=11)
SERVER_DEFAULT = {"n": 2, "nu_1": 9, "nu_2": 5, "p": 256, "q2_bits": 22, "t_gsw": 7, "t_conv": 3, "t_exp_left": 5,
                  "t_exp_right": 5, "instances": 4, "db_item_size": 32768}   # server.rs:1052-1066
SMALL_INST2 = dict(FAST, instances=2, db_item_size=16384)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
