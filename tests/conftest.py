import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# RD_TEST_FULL=1: the fuzz / soak-like GPU tests at their full sizes (500 + 300 fuzzed files, the 2^20-read oracle tail): what the builder runs on
# its own lease; the default sizes keep the driver's GPU suite under seven minutes
FULL = os.environ.get("RD_TEST_FULL") == "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O.load_default()


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np

    class G:
        def npz(self, name):
            return np.load(os.path.join(GOLDEN, name + ".npz"))

        def json(self, name):
            with open(os.path.join(GOLDEN, name + ".json")) as fh:
                return json.load(fh)
    return G()


@pytest.fixture(scope="session")
def gpu_model():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ribodetector_amd.model import model as module_arch
    from ribodetector_amd.parse_config import ConfigParser
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    m = cfg.init_obj("arch", module_arch)
    m.load_state_dict(cfg.load_state_dict("mcc"))
    m.to("cuda:0").eval()
    m.set_prefix_table("auto")       # the configuration the CLI and bench.py run (the table is opt-in for a bare SeqModel)
    return m


@pytest.fixture(scope="session")
def report():
    """numbers worth keeping from a GPU test session (max logit errors ...): written to gpurun_out/parity_report.json"""
    import json
    data = {}
    yield data
    if data:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
