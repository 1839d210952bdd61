"""pytest plugin of the reference-suite subprocess: the device new tensors default to (REFSUITE_DEVICE)."""
import os

import torch


def pytest_configure(config):
    dev = os.environ.get("REFSUITE_DEVICE", "cpu")
    if dev != "cpu":
        torch.set_default_device(dev)
