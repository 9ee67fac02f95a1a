"""`python3 bench.py --gpus N` starts its own rank processes (the form the driver uses for N = 1 must also work for N > 1 without an
external launcher).  On the GPU box: two ranks rehearsed on the one card -- gloo on the host side, mailboxes from host-exchanged IPC
handles, the folded exchange in every iteration -- and the line must say what ran.  Without a GPU: the parent must come back with a
non-zero exit code instead of hanging or printing a line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_without_gpu_fails_loudly():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu test below")
    p = _bench("--gpus", "2", "--docs", "50", "--launch-timeout", "240", timeout=300)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "no GPU visible" in p.stderr


@pytest.mark.gpu
def test_two_self_launched_ranks_on_one_card():
    p = _bench("--gpus", "2", "--docs", "2000", "--steps", "10", "--warmup", "2", "--repeats", "3", "--launch-timeout", "600")
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    c = r["config"]
    assert r["n_gpus"] == 2 and c["comm_nranks"] == 2 and c["comm_nranks_per_rank"] == [2, 2]
    assert c["allreduce_per_rank"] == ["p2p", "p2p"] and c["docs_per_rank"] == [2000, 2000] and c["docs_total"] == 4000
    assert r["value"] > 0 and abs(r["value"] - 4000 / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]
    assert "also" not in r          # --docs given: only the one configuration
    # strong scaling: the one corpus split by nonzeros
    p = _bench("--gpus", "2", "--docs", "2000", "--steps", "10", "--warmup", "2", "--repeats", "3", "--scaling", "strong")
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert sum(r["config"]["docs_per_rank"]) == 2000 and r["config"]["docs_total"] == 2000 and "note" in r["config"]


@pytest.mark.gpu
def test_default_invocation_with_two_ranks_carries_every_configuration():
    """The form the driver's scaling runs use: `bench.py --gpus N` with nothing else, i.e. config 2 and then, on the same ranks and the
    same communicator, configs 4, 5 and the 640k-document LDA under "also".  Every library call that ends in the ranks' exchange must
    be made by every rank: round 3 shipped for a while with the ll history read by rank 0 alone (it flushes the lagged log-likelihood,
    an all-reduce), which left rank 0 one exchange ahead -- harmless at the end of a run, a chain of 20-second time-outs before the
    next configuration."""
    p = _bench("--gpus", "2", "--launch-timeout", "600", timeout=700)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["config"]["comm_nranks_per_rank"] == [2, 2]
    assert set(r["also"]) == {"cfg4", "cfg5", "lda_640k_docs"}
    for k, v in r["also"].items():
        assert "error" not in v and v["value"] > 0 and v["n_gpus"] == 2, (k, v.get("error"))
    assert r["ll_last"] < 0 and r["also"]["lda_640k_docs"]["ll_last"] < 0
    # what the N > 1 line must carry: the headline weak AND strong; configs 4 / 5 on THEIR corpus (50k / 100k documents in
    # total) sharded over the ranks with the weak variant beside; every entry per transport, with the ranks' view of the communicator
    assert r["scaling"] == "weak" and r["config"]["docs_total"] == 20000
    st = r["strong"]
    assert st["scaling"] == "strong" and st["config"]["docs_total"] == 10000 and sum(st["config"]["docs_per_rank"]) == 10000 and "note" in st["config"]
    want = {"cfg4": 50000, "cfg5": 100000}
    for k, total in want.items():
        e = r["also"][k]
        assert e["scaling"] == "strong" and e["config"]["docs_total"] == total and sum(e["config"]["docs_per_rank"]) == total, k
        assert e["weak"]["scaling"] == "weak" and e["weak"]["config"]["docs_total"] == 2 * total and e["weak"]["value"] > 0, k
    for e in (r, st, r["also"]["cfg4"], r["also"]["cfg5"]):
        c = e["config"]
        assert c["comm_nranks_per_rank"] == [2, 2] and c["allreduce_per_rank"] == ["p2p", "p2p"] and len(c["docs_per_rank"]) == 2
        assert 0 < e["roofline"]["frac"] < 1 and 0 < e["roofline"].get("hbm", e["roofline"])["frac"] < 1
        t = e["transports"]
        assert set(t) == {"p2p", "rccl"} and t["p2p"]["value"] > 0 and t["p2p"]["allreduce_per_rank"] == ["p2p", "p2p"] and 0 < t["p2p"]["roofline_frac_per_gpu"] < 1
        # two ranks on ONE card: RCCL refuses that, and the line must say so rather than carry a number that did not run
        assert "unavailable" in t["rccl"] or (t["rccl"]["value"] > 0 and t["rccl"]["allreduce_per_rank"] == ["rccl", "rccl"])


@pytest.mark.gpu
def test_single_gpu_line_keeps_its_keys():
    p = _bench("--docs", "2000", "--steps", "10", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline")
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "iteration", "elbo_rel_err_vs_oracle"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["config"]["comm_nranks"] == 1 and r["config"]["allreduce"] == "none"
    rf = r["roofline"]
    assert rf["algorithmic_bytes_as_implemented"] <= rf["algorithmic_bytes_per_launch"]
    assert 0 < r["iteration"]["kernel_fraction_of_step"] < 1.5
    assert r["elbo_rel_err_vs_oracle"] < 1e-5


@pytest.mark.gpu
def test_transport_switch_rehearsed_on_one_rank():
    """N > 1 on real hardware measures every entry once over the xGMI mailboxes and once over ncclAllReduce on the same communicator
    (mmm_p2p_enable).  One card cannot hold two RCCL ranks, so the switch is rehearsed on a ONE-rank communicator with mailboxes
    (MMM_FORCE_RCCL=1: every all-reduce really calls RCCL; MMM_P2P_ONE_RANK=1: the mailboxes are set up anyway): both entries must
    have run, on the transport they name, with the same log-likelihood."""
    env = dict(os.environ, MMM_FORCE_RCCL="1", MMM_P2P_ONE_RANK="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--docs", "3000", "--steps", "10", "--warmup", "2", "--repeats", "3", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    t = r["transports"]
    assert r["config"]["allreduce"] == "p2p" and set(t) == {"p2p", "rccl"}
    assert t["p2p"]["allreduce_per_rank"] == ["p2p"] and t["rccl"]["allreduce_per_rank"] == ["rccl"]
    assert t["p2p"]["value"] > 0 and t["rccl"]["value"] > 0
