"""The sharded exchange (xgmi_exchange.XgmiShardStore) over RCCL: the same worker tests/test_distributed_cpu.py runs under
gloo, with the blobs in HBM and the process group on the "nccl" (= RCCL) backend.

World size 1 runs on any GPU box (every collective and the device-side record gather execute; no peer traffic);
world size 2 needs two GPUs -- RCCL refuses two ranks on one device -- and is skipped on a one-GPU box.
"""
import os

import pytest
import torch

from tests.test_distributed_cpu import _exchange_worker, _run_world

pytestmark = pytest.mark.gpu


def test_shard_exchange_rccl_world1():
    assert _run_world(_exchange_worker, 1, 35533 + os.getpid() % 2000, "nccl") == [0]


def test_shard_exchange_rccl_two_devices():
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank; this box has %d" % torch.cuda.device_count())
    assert _run_world(_exchange_worker, 2, 37533 + os.getpid() % 2000, "nccl") == [0, 1]
