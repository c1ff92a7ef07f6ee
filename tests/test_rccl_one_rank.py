"""The data-parallel exchange on the hardware one GPU offers: a one-rank RCCL process group with
FAC_FORCE_ALLREDUCE=1 runs every per-key asynchronous all-reduce(AVG) of the training step -- launched from the gradient hooks
-- and must leave losses and parameters bit-identical to the run without any collective."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(force, native=False):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FAC_FORCE_ALLREDUCE", "FAC_NATIVE_RCCL"):
        env.pop(k, None)
    if native:
        env["FAC_NATIVE_RCCL"] = "1"
    if force:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FAC_FORCE_ALLREDUCE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "rccl_one_rank.py")], env=env, capture_output=True, text=True,
                       timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]      # RCCL prints its banner on stdout too
    return json.loads(line)


@pytest.mark.gpu
def test_one_rank_rccl_exchange_is_the_identity():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    plain, forced = _run(False), _run(True)
    assert plain["process_group"] is False and forced["process_group"] == "nccl"
    assert forced["losses"] == plain["losses"]
    assert forced["param_sums"] == plain["param_sums"]
    assert all(v == "none" for rep in plain["exchange_launched_from"] for v in rep.values())
    # every iteration, the first one included (launch points are fixed by the graph, not learnt from a previous step): the decoder's
    # and the quantizer's buckets leave from the hooks on the decoder input's / the latent's gradient, the encoder's later buckets
    # from its block-boundary hooks; the discriminator's after its own backward
    for rep in forced["exchange_launched_from"]:
        assert rep["decoder"] == "hook" and rep["quantizer"] == "hook" and rep["encoder"] == "hook" and rep["discriminator"] == "end", rep
    last = forced["bucket_launches"][-1]
    assert len(last["decoder"]) >= 5 and all(w == "hook" for _, w in last["decoder"])           # 342 MB in <= 64 MB buckets
    assert [b for b, _ in last["encoder"]] == list(range(len(last["encoder"])))                  # fixed order, end of the arena first
    assert last["encoder"][-1][1] == "end" and last["encoder"][0][1] == "hook"                  # only the remainder waits for the end
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(dict(plain=plain, forced=forced), open(os.path.join(REPO, "gpurun_out", "rccl_one_rank.json"), "w"), indent=1)


@pytest.mark.gpu
def test_one_rank_native_rccl_arena_exchange_is_the_identity():
    """The same with FAC_NATIVE_RCCL=1: every bucket goes through `fac_allreduce_arena` (include/facodec_hip.h: ncclAllReduce(ncclAvg)
    behind the C ABI, on the exchange stream facodec_amd.optim.NativeRccl owns, communicator made by fac_rccl_comm_init from an id
    of fac_rccl_unique_id) instead of torch.distributed's all_reduce -- same launch points, bit-identical losses and parameters."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    plain, native = _run(False), _run(True, native=True)
    assert plain["native_rccl_calls"] is None and native["native_rccl_calls"] > 3 * 10       # >= 10 buckets + 4 flag words, 3 steps
    assert native["losses"] == plain["losses"] and native["param_sums"] == plain["param_sums"]
    for rep in native["exchange_launched_from"]:
        assert rep["decoder"] == "hook" and rep["quantizer"] == "hook" and rep["encoder"] == "hook" and rep["discriminator"] == "end", rep


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_over_gloo_with_asymmetric_masks():
    """The closest thing to a second GPU on a one-GPU box: two processes run the whole train step (real kernels, side streams,
    gradient hooks) on the same device and exchange over gloo; odd ranks drop their residual quantizers, so the ranks reach
    different parameter sets.  tools/ddp_smoke.py asserts identical parameters on both ranks and the same bucket launch sequence."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, FAC_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FAC_FORCE_ALLREDUCE"):
        env.pop(k, None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "tools", "ddp_smoke.py"), "--predictors"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    rep = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rep["world"] == 2 and rep["params_identical_across_ranks"] and rep["losses_finite"] and rep["with_predictors"]
    assert all(w.endswith(":hook") for w in rep["bucket_launches"]["decoder"] + rep["bucket_launches"]["fa_predictors"])
