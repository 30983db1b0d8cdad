"""GPU: bench.py honours its output contract (one JSON line with the driver's keys, the roofline
and cpu_baseline objects) on the small workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_contract_small_workload():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--workload", "mini", "--cpu-dets", "4", "--pcg-extra"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "det-samples/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["config"]["workload"] == "mini"
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_exact"):
        assert key in r, key
    assert r["traffic"] is None and r["traffic_exact"] is None       # (no committed PMC profile of the mini workload)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the two kernels' own byte mixes as plain streams over the same arrays (k_probe_byte_mix) ride along with the line
    sc = r["stream_ceiling"]
    for key in ("read_write_GBs", "scan_map_byte_mix_ms", "build_noise_weighted_byte_mix_ms", "scan_map_frac_of_its_mix"):
        assert sc[key] > 0, key
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    assert c["one_thread"] > 0 and c["best_threads"]["value"] >= max(c["thread_sweep"].values()) * (1 - 1e-12)
    assert "1" in c["thread_sweep"] and str(c["cores"]) in c["thread_sweep"]
    # the reported value is the best point of the sweep with the threads it used; the all-core point is kept beside it
    assert c["value"] == c["best_threads"]["value"] and c["cores"] == c["best_threads"]["threads"]
    assert str(c["all_cores"]["threads"]) in c["thread_sweep"]
    # the buffers are allocated like the operators allocate them; the FFT noise weighting is reported separately
    assert "toast_hip::Manager" in d["allocator"] and d["placement"] is None
    assert {"slabs", "slab_mallocs", "malloc_ms", "max_malloc_ms", "slab_GB", "peak_used_GB", "direct_mallocs",
            "interleaved_slabs", "chunks", "chunks_other_zone", "chunks_other_wanted", "placement_ok", "search_exhausted",
            "probes", "probes_by_clock", "searches", "searches_capped_ms"} <= set(d["allocator_stats"])
    # the zone search's passes are timed on the device, and the line says whether the placement worked out
    assert d["allocator_stats"]["probes_by_clock"] == d["allocator_stats"]["probes"]
    assert isinstance(d["allocator_stats"]["placement_ok"], bool) and d["allocator_stats"]["searches_capped_ms"] == 0
    if os.environ.get("TOAST_HIP_ALLOC", "") == "plain":        # (the switch that turns the arena off)
        assert d["allocator_stats"]["direct_mallocs"] > 0 and d["allocator_stats"]["slab_mallocs"] == 0
    else:
        assert d["allocator_stats"]["direct_mallocs"] == 0 and d["allocator_stats"]["slab_mallocs"] >= 1
    f = d["fft_noise_weight"]
    assert f["ms"] > 0 and f["samples_per_s"] > 0 and f["n_fft"] == 131072 and f["implementation"] == "fused-3pass"
    assert abs(f["pipeline_bytes_per_sample"] - (16 + 32 * 131072 / 50000)) < 1e-9
    # "ms": calls back to back behind an un-timed call; the idle-queue number of rounds 2-4 and the host preparation beside it
    assert f["timing"].startswith("3 calls back to back") and f["ms_from_idle_queue"] > 0 and f["host_prep_ms"] > 0
    assert "long" not in f                                    # (the 512 x 2 880 000 shape rides along with cfg3 only)
    assert d["allreduce"]["bytes"] == 0 and d["kernel_ms"]["allreduce"] >= 0
    # value == whole-job units / time
    n = d["config"]["detectors_per_gpu"] * d["config"]["samples_per_detector"]
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # the fused / on-the-fly variants agree with the operator sequence
    assert d["pcg_lhs_offset_templates"]["fused_vs_sequence_max_rel_diff"] < 1e-12
    assert d["pcg_lhs_offset_templates"]["packed_vs_sequence_max_rel_diff"] < 1e-12
    assert d["pcg_lhs_offset_templates"]["packed_bytes_per_det_sample_and_sweep"] in (18, 20)
    assert d["pointing_on_the_fly"]["offset_lhs_vs_sequence_max_rel_diff"] < 1e-12
    assert d["compact_pixels_weights_on_the_fly"]["offset_lhs_vs_sequence_max_rel_diff"] < 1e-12


def test_bench_two_ranks_code_path():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one rank per GPU), here
    with both ranks sharing the one GPU and gloo instead of RCCL (TOAST_BENCH_SHARE_GPU=1): rank 0
    prints one JSON line, value is the whole-job aggregate, the zmap all-reduce is in the step."""
    env = dict(os.environ, TOAST_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--workload", "mini", "--shard-workload", "cfg2", "--no-fft"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    n = d["config"]["detectors_per_gpu"] * d["config"]["samples_per_detector"]
    assert abs(d["value"] - 2 * n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only
    # the all-reduce of the zmap is timed and sized; the extra shard workload (what --gpus 8 does with cfg4) is attached
    assert d["allreduce"]["bytes"] == d["config"]["n_local_submap"] * 3072 * 3 * 8 and d["allreduce"]["ms"] > 0
    assert d["kernel_ms"]["allreduce"] > 0 and d["allreduce"]["backend"] == "gloo"
    sh = d["configs3_shard"]
    assert sh["config"]["workload"] == "cfg2" and sh["value"] > 0 and sh["allreduce"]["bytes"] > 0


def test_bench_self_launch_two_ranks():
    """Exactly the driver's command form `python3 bench.py --gpus 2 ...` (no launcher, no RANK in the
    environment): bench.py starts torch.distributed.run itself as a child process and relays rank 0's one
    JSON line.  Both ranks share the one GPU here (TOAST_BENCH_SHARE_GPU=1, gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                             "MASTER_PORT")}
    env.update(TOAST_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup",
                          "1", "--workload", "mini", "--no-fft"], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["allreduce"]["bytes"] > 0 and d["allreduce"]["ms"] > 0
    n = d["config"]["detectors_per_gpu"] * d["config"]["samples_per_detector"]
    assert abs(d["value"] - 2 * n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_bench_collective_path_with_one_rank():
    """The N > 1 code path of bench.py on a single-GPU box (TOAST_BENCH_SINGLE_RANK_COMM=1: a one-rank `nccl` process
    group): the ranks agree that RCCL can be loaded, the library's communicator is created from the id that the process
    group carries, its all-reduce and owner-computes pass are checked against torch.distributed, and the timed step
    reduces the map through it on the kernels' stream."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(TOAST_BENCH_SINGLE_RANK_COMM="1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload",
                          "mini", "--no-fft", "--no-cpu-baseline", "--shard-workload", "cfg2"], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert len(out.stdout.strip().splitlines()) == 1, out.stdout[:600]     # nothing but the JSON line on stdout
    d = json.loads(out.stdout.strip())
    a = d["allreduce"]
    assert d["n_gpus"] == 1 and a["backend"] == "nccl" and a["note"] is None, a
    assert a["implementation"].startswith("toast_hip_comm") and a["bytes"] > 0 and a["ms"] > 0
    assert a["owner_computes_reduce_apply_ms"] > 0
    sh = d["configs3_shard"]["allreduce"]           # a second workload in the same process reuses the communicator
    assert sh["implementation"].startswith("toast_hip_comm") and sh["note"] is None
