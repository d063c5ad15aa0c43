"""bench.py without a GPU: the synthetic-input generator, the self-spawn of `--gpus N`, the multi-rank timing fences and the
single JSON line -- run at world 2 over gloo on the host emulation of the kernels (tiny shape; not a measurement)."""
import hashlib
import importlib.util
import json
import os
import subprocess
import sys

import torch

from oracle import crossclr_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_inputs_are_the_oracle_generator():
    b = _bench_module()
    for seed in (1234, 1235):
        v1, t1 = b.make_inputs(96, 40, seed)
        v2, t2 = orc.make_inputs("randn", 96, 40, seed)
        assert torch.equal(v1, v2) and torch.equal(t1, t2)


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must be exactly one JSON line, got: {r.stdout!r}"
    return json.loads(lines[0])


def test_bench_single_rank_line_on_the_emulated_kernels():
    out = _run(["--selftest-emu", "--rows", "32", "--dim", "32", "--steps", "2", "--warmup", "1", "--prewarm", "0", "--mode", "fp32"])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["unit"] == "pairs/s"
    assert out["value"] > 0 and out["higher_is_better"] is True and out["scaling"] == "weak" and out["vs_baseline"] is None
    v, t = orc.make_inputs("randn", 32, 32, 1234)
    assert abs(out["loss"] - float(orc.streaming_stats(v, t, 0.03, 0.8)["loss"])) < 1e-4


def test_bench_gpus2_spawns_itself_and_reports_the_global_loss():
    """`python bench.py --gpus 2` with no launcher in the environment: re-execs under torch.distributed.run, two gloo ranks
    (emulated kernels), max-over-ranks timing, rank 0 prints the line, clean teardown."""
    out = _run(["--gpus", "2", "--selftest-emu", "--rows", "32", "--dim", "32", "--steps", "2", "--warmup", "1", "--prewarm", "0",
                "--mode", "fp32"])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64
    v0, t0 = orc.make_inputs("randn", 32, 32, 1234)
    v1, t1 = orc.make_inputs("randn", 32, 32, 1235)
    ref = float(orc.streaming_stats(torch.cat([v0, v1]), torch.cat([t0, t1]), 0.03, 0.8)["loss"])
    assert abs(out["loss"] - ref) < 1e-4


def test_bench_gpus8_selftest_with_the_per_peer_exchange():
    """`bench.py --gpus 8 --selftest-emu` (BASELINE configs 4 / 5's world size): eight gloo ranks, the pair scheme with three pair
    partners + the antipodal rank, operands exchanged peer by peer (CROSSCLR_EXCHANGE=p2p_each: one launch per partner as its slice
    lands); the line carries the per-rank diagnostics block and names the exchange."""
    out = _run(["--gpus", "8", "--selftest-emu", "--rows", "8", "--dim", "16", "--steps", "1", "--warmup", "0", "--prewarm", "0",
                "--mode", "bf16"], env_extra={"CROSSCLR_EXCHANGE": "p2p_each"}, timeout=1800)
    assert out["n_gpus"] == 8 and out["config"]["global_batch"] == 64
    assert len(out["per_rank"]) == 8 and {r["exchange"] for r in out["per_rank"]} == {"p2p_each"}
    vs, ts = zip(*[orc.make_inputs("randn", 8, 16, 1234 + r) for r in range(8)])
    ref = float(orc.bf16_operand_model_loss(torch.cat(vs), torch.cat(ts), 0.03, 0.8))
    assert abs(out["loss"] - ref) < 5e-3 * max(1.0, abs(ref))


def test_bench_gpus8_selftest_measures_the_three_exchanges():
    """Without CROSSCLR_EXCHANGE the multi-rank bench runs its timed region once per way the operands can travel (all-gather, batched
    point-to-point, per peer) and reports the table; `value` / `ms_per_step` are the fastest form's, which the per-rank block names."""
    out = _run(["--gpus", "8", "--selftest-emu", "--rows", "8", "--dim", "16", "--steps", "1", "--warmup", "0", "--prewarm", "0",
                "--mode", "bf16"], timeout=2400)
    px = out["per_exchange"]
    assert set(px) == {"allgather", "p2p", "p2p_each"} and sum(1 for v in px.values() if v["winner"]) == 1
    best = next(k for k, v in px.items() if v["winner"])
    assert abs(px[best]["ms_per_step"] - out["ms_per_step"]) <= 1e-3 * out["ms_per_step"] + 1e-4
    assert all(px[best]["ms_per_step"] <= v["ms_per_step"] for v in px.values())
    assert {r["exchange"] for r in out["per_rank"]} == {best}
    vs, ts = zip(*[orc.make_inputs("randn", 8, 16, 1234 + r) for r in range(8)])
    ref = float(orc.bf16_operand_model_loss(torch.cat(vs), torch.cat(ts), 0.03, 0.8))
    assert abs(out["loss"] - ref) < 5e-3 * max(1.0, abs(ref))


def test_a_failing_exchange_form_still_yields_a_line():
    """N >= 3 runs the three operand-exchange forms back to back; if one of the later ones raises on a rank (its peers then hang in
    their collectives) every rank leaves through the watchdog and rank 0 prints the line of the forms that finished."""
    out = _run(["--gpus", "3", "--selftest-emu", "--rows", "8", "--dim", "16", "--steps", "1", "--warmup", "0", "--prewarm", "0",
                "--mode", "bf16", "--exchange-deadline", "30"], env_extra={"CROSSCLR_BENCH_INJECT_FAILURE": "p2p"}, timeout=900)
    assert out["n_gpus"] == 3 and out["config"]["operand_exchange"] == "allgather"
    assert out["per_exchange"]["allgather"]["winner"] and "error" in out["per_exchange"]["p2p"]
    assert out["value"] > 0 and out["config"]["global_batch"] == 24
