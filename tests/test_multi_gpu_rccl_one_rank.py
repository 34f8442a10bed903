"""The multi-GPU code on REAL device memory with the one GPU a test box has: torch.distributed
with backend "nccl" (= RCCL) in a world of ONE rank — RCCL initialisation, shard.broadcast_key on device tensors into a key-switch
plan, shard.allgather_words (scatter + all-gather) of a packed bootstrapping key set on HIP memory, hal/bootstrap_batch.cpp's
fbb_export_keys -> fbb_adopt_keys of windows of a torch CUDA tensor, and a batch of bootstraps on the ADOPTED keys whose outputs
are byte-identical to the same batch bootstrapped on the locally generated keys.  The 2-rank forms of the same calls run on gloo +
the lane emulator in tests/test_multi_gpu_gloo.py; the hardware scaling curve is the driver's 8-GPU run."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB = os.path.join(ROOT, "openfhe-development_amd", "csrc", "libfhe_hip.so")
PRNG = os.path.join(ROOT, "tests", "hal", "_build", "libdetprng.so")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               FHE_HIP_LIB=HIP_LIB, FHE_HAL_REQUIRE_DEVICE="1", OMP_NUM_THREADS="4")
    env.pop("FHE_HAL_ALLOW_HOST", None)
    return env


WORKER = r'''
import os, sys
import numpy as np
ROOT, out, prng, mode = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist
from openfhe_amd import fhe_hip as fh
from openfhe_amd import shard, boot_batch as bb
import libs
if mode == "local":  # the same batch on locally generated keys, a process of its own (the deterministic PRNG is per process)
    r2 = bb.run_rank(12, 8, 3, 2, 1, 0, prng, dump_path=out + ".local.bin", warmup=0, key_threads=2)
    r2.pop("handle").close()
    sys.exit(0)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
# (1) the evaluation key of a key-switch plan arrives through the broadcast into a DEVICE tensor the plan adopts; EvalMult vs the oracle
lib = fh.Lib()
o = libs.load_oracle()
logN, sizeQ, dnum, B = 12, 4, 2, 3
N = 1 << logN
q, psiQ = lib.ckks_like_chain(logN, sizeQ, 60, 50)
p, psiP = lib.select_p(logN, q, dnum)
allq = np.concatenate([q, p])
ctx = fh.Context(lib, logN, allq, np.concatenate([psiQ, psiP]))
plan = fh.KeySwitchPlan(ctx, sizeQ, len(p), dnum)
rng = np.random.default_rng(5)
keyB, keyA = libs.rand_tower(rng, allq, N, dnum), libs.rand_tower(rng, allq, N, dnum)
ops = [libs.rand_tower(rng, q, N, B) for _ in range(4)]
keep = shard.broadcast_key(plan, keyB, keyA, dev)
assert keep[0].is_cuda
r0, r1 = plan.EvalMult(*(ctx.tower(x) for x in ops))
hy = o.orc_hybrid_create(N, sizeQ, q, psiQ, len(p), p, psiP, dnum)
for b in range(B):
    c0, c1 = np.empty_like(ops[0][b]), np.empty_like(ops[0][b])
    o.orc_ckks_eval_mult_relin(hy, ops[0][b], ops[1][b], ops[2][b], ops[3][b], sizeQ, keyB, keyA, c0, c1)
    assert np.array_equal(r0.to_host()[b], c0) and np.array_equal(r1.to_host()[b], c1), "EvalMult with the broadcast key differs from the oracle"
print("broadcast_key on device tensors: EvalMult bit-exact vs the oracle")
# (2) scatter + all-gather of words on device memory: the gathered tensor equals the source
src = torch.arange(3 * 5 * 1000 + 7, dtype=torch.int64, device=dev) * 2654435761
got = shard.allgather_words(src, (3 * 5 * 1000 + 7,), dev, src=0)
assert got.is_cuda and torch.equal(got, src)
print("allgather_words on device memory: identical")
plan.close(); ctx.close()
# (3) the bootstrapping key set exported, gathered and ADOPTED as windows of the gathered CUDA tensor; the batch bootstrapped on them
r = bb.run_rank(12, 8, 3, 2, 1, 0, prng, dist=dist, torch_device=dev, dump_path=out + ".adopted.bin", warmup=0, key_threads=2,
                force_replication=True)
assert r["key_set_GB"] > 0 and r["keys"].is_cuda, r
print("replicated key set GB", r["key_set_GB"], "GB/s", r["key_replication_GBps"], "errors", r["max_abs_error"])
r.pop("handle").close()
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_rccl_one_rank_key_replication_and_adopted_key_windows(tmp_path):
    sys.path.insert(0, ROOT)
    from openfhe_amd import boot_batch as bb
    if not (os.path.exists(bb.HIP_SO) and os.path.exists(PRNG)):
        pytest.skip("hal/_build/libfhe_boot_batch_hip.so not built (./build.sh hal needs the reference sources)")
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = str(tmp_path / "boot")
    p = subprocess.run([sys.executable, str(worker), ROOT, out, PRNG, "adopted"], env=rank_env(), capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    p2 = subprocess.run([sys.executable, str(worker), ROOT, out, PRNG, "local"], env=rank_env(), capture_output=True, text=True, timeout=1200)
    assert p2.returncode == 0, p2.stdout[-3000:] + p2.stderr[-3000:]
    assert "EvalMult bit-exact" in p.stdout and "allgather_words on device memory: identical" in p.stdout
    a, b = open(out + ".adopted.bin", "rb").read(), open(out + ".local.bin", "rb").read()
    assert len(a) > 10000 and a == b, "bootstraps on the adopted (replicated) keys differ from those on the locally generated keys"


@pytest.mark.gpu
def test_bench_forced_dist_on_one_gpu(tmp_path):
    """bench.py with FHE_BENCH_FORCE_DIST=1: backend nccl, one rank — the bench's own distributed path (key broadcast of the EvalMult leg,
    the rotation-key replication leg, the sharded bootstrap leg with adopted key windows and the lockstep-vs-narrow comparison)"""
    if not os.path.exists(PRNG):
        pytest.skip("tests/hal/_build not built")
    env = rank_env()
    env["FHE_BENCH_FORCE_DIST"] = "1"
    env["FHE_BENCH_KEYREP_KEYS"] = "3"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--steps", "1", "--warmup", "0", "--evalmult-batch", "4",
                        "--no-hadamard", "--no-bfv", "--no-lt", "--no-cc-evalmult", "--no-cpu-baseline", "--bootstrap-logn", "13",
                        "--bootstrap-batch", "4", "--bootstrap-threads", "2"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads(p.stdout.strip().split("\n")[-1])
    assert line["n_gpus"] == 1 and "rccl" in line["evalmult"]["eval_key"], line["evalmult"]["eval_key"]
    assert line["rotation_key_replication"]["GB"] > 0, line.get("rotation_key_replication")
    boot = line["evalbootstrap"]
    assert "error" not in boot and boot["key_set_GB"] > 0 and boot["key_replication_GBps"] > 0, boot
    assert boot["lockstep"]["parity"].startswith("all 4 outputs identical"), boot["lockstep"]
    assert boot["max_abs_error_vs_message"] < 1e-3
