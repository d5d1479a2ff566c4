import json
import os
import sys
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def crc(t) -> int:
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_json(name: str):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def state_dict():
    from isp_tts_amd.synth import make_state_dict
    return make_state_dict()


@pytest.fixture(scope="session")
def gpu_model(state_dict):
    """The product model on cuda:0 with the synthetic weights.  Fails (does not skip) without the HIP library."""
    from isp_tts_amd import runtime
    from isp_tts_amd.acoustic import AcousticModel
    from isp_tts_amd.config import AcousticDims
    runtime.lib()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    model = AcousticModel.init(AcousticDims().model_config()).eval()
    model.load_state_dict(state_dict, strict=True)
    # frozen: the inference kernels.  (With trainable parameters and gradients enabled `model(...)` is the training forward -
    # tests/test_gpu_train.py builds its own models for that.)
    return model.to("cuda").requires_grad_(False)
