"""Sanity of XgmiShardStore on the RCCL backend with a single rank (the multi-rank logic is covered by the
gloo tests; this only checks that the device-tensor / nccl control path runs on the GPU box)."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from lmcache_amd.storage_backend.connector.xgmi_exchange import XgmiShardStore
st = XgmiShardStore()
print("store ok", st.device, flush=True)
assert st.device.type == "cuda"
blobs = {f"k{i}": torch.randint(0, 256, (1000 + 37 * i,), dtype=torch.uint8, device="cuda") for i in range(5)}
assert st.exchange_put(list(blobs.items())) == 5
print("put ok", flush=True)
got = st.exchange_get(["k3", "missing", "k0"])
print("get ok", flush=True)
assert got[1] is None and torch.equal(got[0], blobs["k3"]) and torch.equal(got[2], blobs["k0"]) and got[0].is_cuda
t = torch.ones(1, device="cuda"); dist.all_reduce(t); dist.barrier()
dist.destroy_process_group()
print("exchange on nccl world=1: ok")
