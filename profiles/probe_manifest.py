"""What profiles/pmc_probe.py launches, in order -- read by the probe (REPS) and by profiles/collect_pmc.py, which asserts that the
dispatch sequence rocprofv3 recorded is exactly this one before attributing counters to tags."""
REPS = 3        # every group is launched this many times in a row
# the probe's launches in order: groups of (bench.py tag, kernel-name substrings of one launch of the op); ops that share a
# kernel symbol (the pixel gather of both sample adjoints) are told apart by their position in the dispatch sequence
PROBE_GROUPS = [
    [("t2h_segmean_fwd[C=512,r=32]", ["segmean_cells_kernel", "segmean_finalize_kernel"])],
    [("t2h_sample_fwd[C=512,r=32]", ["sample_fwd_kernel"]),
     ("t2h_sample_bwd[C=512,r=32]", ["sample_bwd_cells_", "sample_bwd_gather9_kernel"])],
    # r03, deferred point update: per-cell sums of the widest hidden activations at the finest resolution
    [("t2h_segsum_fwd[C=1024,r=256]", ["segmean_fwd_kernel<4, false>"])],
    [("t2h_segsum_bwd_multi[C=1024,n=4]", ["segsum_bwd_multi_kernel"])],
    # r03, hidden activations on chip: sample + ReLU + per-cell sums + sign bits; the backward walk + its pixel gather
    [("t2h_cell_order_build[r=32]", ["cell_order_kernel"])],
    [("t2h_sample_relu_cellsums[C=1024,r=32]", ["sample_relu_cellsums_v2_kernel"])],
    [("t2h_sample_bwd_from_sums[C=1024,r=32]", ["sample_bwd_walk_kernel", "sample_bwd_gather9_kernel"])],
]


