"""Host logic of bench.py (no GPU): the per-kernel aggregation and the size of the printed line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _FakeTimeline:
    def __init__(self, rows):
        self.rows = rows

    def summary(self):
        return self.rows


def _rows():
    rows = {}
    for i in range(130):      # more entry-point shapes than a real step has
        sym = "gemm_dma_kernel" if i % 3 == 0 else ("gemm_dma_tn_kernel" if i % 3 == 1 else f"t2h_op_{i}")
        rows[f"t2h_linear_fwd[K={i},N={2 * i}]"] = {"calls": 4, "ms": 0.4 + 0.01 * i, "bytes": 10 ** 6 * (i + 1),
                                                    "flops": (10 ** 11) * (i % 3 != 2), "symbol": sym}
    return rows


def test_symbol_aggregation_is_a_class_total():
    tags, syms = bench.kernel_tables(_FakeTimeline(_rows()), n_steps=2)
    assert len(tags) == 130 and syms[0]["kernel"] in ("gemm_dma_kernel", "gemm_dma_tn_kernel")
    top = syms[0]
    members = [t for t in tags if t["symbol"] == top["kernel"]]
    assert abs(top["ms_per_step"] - sum(m["ms_per_step"] for m in members)) < 1e-3 * len(members)
    assert top["launches_per_step"] == sum(m["launches_per_step"] for m in members)
    # achieved = total flops / total time, not the best member
    total_f = sum(m["flops_per_launch"] * m["launches_per_step"] for m in members)
    total_s = sum(m["ms_per_step"] for m in members) * 1e-3
    assert abs(top["TFLOPs"] - total_f / total_s / 1e12) <= 0.02 * top["TFLOPs"]


def test_printed_line_stays_compact():
    tags, syms = bench.kernel_tables(_FakeTimeline(_rows()), n_steps=2)
    out = {"metric": "training tiles/sec (Berlin crop, cloud-only)", "value": 41.0, "unit": "tiles/s", "n_gpus": 1,
           "steps": 20, "warmup": 5, "ms_per_step": 24.4, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "config": {"workload": "x" * 160},
           "roofline": bench.roof(syms[0]), "roofline_scatter_reduce": [bench.roof(t) for t in tags[:5]],
           "roofline_top_symbols": [{"kernel": s["kernel"][:60], "ms_per_step": 1.0, "frac": 0.5, "bound": "mfma"} for s in syms[:6]],
           "cpu_baseline": {"value": 0.05, "unit": "tiles/s", "cores": 8, "kind": "port", "sample": "y" * 200,
                            "all_cores": {"value": 0.05, "cores": 128, "median_s": 20.0, "timed_steps": 3}}}
    assert len(json.dumps(out)) < 4096


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` the way the driver calls it: the ranks must be started as a child torch.distributed.run
    (here, without a GPU, each rank then refuses to run -- which proves they were started, with the command line
    passed through, and that their exit code is propagated)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1" in r.stderr
    assert "--gpus 2 --steps 2 --warmup 1" in r.stderr
    assert r.stderr.count("no GPU visible") == 2 and "must be launched with" not in r.stderr
    assert r.stdout.strip() == ""


def test_mismatched_world_size_is_refused():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4 does not match --gpus 2" in r.stderr
