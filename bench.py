#!/usr/bin/env python3
"""Benchmark of the Verlet-list build (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one neighbour-list build over positions already resident in HBM (the reference's timing loop,
make_list.cu:124-127: MakeNeighList(q, N, sync=false) back to back, one device sync at the end).

  N = 1   workload = BASELINE config 2: 1 048 576 particles, rho = 1.0, rc = 3.3, fp32, uniform random box.
  N > 1   weak scaling: N x 1 048 576 particles at the same density, cut into z-slabs of whole cell layers, one
          rank per GPU; every step does the ghost-layer exchange (point-to-point over RCCL/xGMI) and the slab
          build.  ``--workload cfg4`` instead fixes the total at 33 554 432 particles (strong scaling, BASELINE
          config 4).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      dominant kernel (the list-filling pair search): algorithmic bytes / launch time vs 8 TB/s HBM
  build         whole-build algorithmic bytes B = N*sizeof(Vec) + 4P + 4(N+1) + 4N over ms_per_step
  cpu_baseline  the reference's own CPU classes (oracle/_ref, compiled from the reference), 1 core, same input
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
RC = 3.3
N_PER_GPU = 1 << 20


def cpu_baseline(q32, box, budget_s=40.0):
    """Times the reference CPU path on this host (rank 0, N = 1 only).  Sample = ONE full build of the same 1M
    workload per variant (a build takes seconds on one core).  kind 'reference' = the reference's classes compiled
    from its sources (oracle/_ref); 'port' = the C restatement when those are absent."""
    from oracle import pyoracle as po

    n = len(q32)
    variants = {}
    t_start = time.time()
    q64 = None
    for name, label in (("fused_native", "scalar NeighList<float3> -DLOOP_FUSION (fp32)"),
                        ("avx512_8x1", "NeighListAVX512 8x1 (fp64 only)"),
                        ("avx2_4x1", "NeighListAVX2 4x1 (fp64 only)")):
        if not po.ref_runnable(name) or time.time() - t_start > budget_s:
            continue
        if name == "fused_native":
            _, secs, npairs = po.ref_build(q32, RC, box, name, loops=1, want_list=False)
        else:
            if q64 is None:
                q64 = q32.astype(np.float64)
            _, secs, npairs = po.ref_build(q64, RC, box, name, loops=1, want_list=False)
        variants[name] = {"what": label, "seconds_per_build": round(secs, 3), "npairs": npairs,
                          "mpairs_per_s": round(npairs / secs / 1e6, 3)}
    if variants:
        best = max(variants.values(), key=lambda v: v["mpairs_per_s"])
        return {"value": best["mpairs_per_s"], "unit": "Mpairs/s", "cores": 1, "kind": "reference",
                "sample": f"1 build of the full workload (N={n}, rho=1.0, rc=3.3) per reference class, single "
                          f"thread; value = fastest class ({best['what']})",
                "variants": variants, "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}
    t0 = time.time()
    h = po.build(q32, RC, box)
    secs = time.time() - t0
    return {"value": round(h.npairs / secs / 1e6, 3), "unit": "Mpairs/s", "cores": 1, "kind": "port",
            "sample": f"1 build of the full workload (N={n}) with the C restatement oracle/nl_oracle.c",
            "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _measured_copy_gbs(torch, dev):
    """Device-to-device copy rate (read + write bytes per second) of this GPU right now: what the reference reports
    next to its numbers as GPU_STREAM_RESULT.txt.  1 GiB buffers, best of 5."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * 4.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return round(best, 1)


def _traffic_from_profiles(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC passes (profiles/*_pmc.json), or None."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_pmc.json"):
                best = os.path.join(pdir, f)
    if not best:
        return None
    try:
        d = json.load(open(best))
        return d.get("hbm_bytes_per_launch", {}).get(kernel)
    except Exception:
        return None


def _reference_pairs(n_total, density, dtype, got, world=1, weak=True):
    """The pair count of this box from the compiled reference (tests/golden/known_answers.json, written by
    oracle/gen_golden.py --big; a committed fixture, nothing under oracle/ runs here): "ok" when the build (the union
    over ranks for N > 1) found exactly that many pairs, the two numbers otherwise, None when no answer is stored."""
    try:
        ka = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    except Exception:
        return None
    if n_total % (1 << 20):
        return None
    rho = "1" if density == 1.0 else "05"
    key = f"u1M_rho{rho}_{dtype}" if world == 1 else f"w{world}x1M_rho{rho}_{dtype}" if weak else f"u{n_total >> 20}M_rho{rho}_{dtype}"
    if key not in ka:
        return None
    want = int(ka[key]["npairs"])
    return "ok" if want == int(got) else {"got": int(got), "reference": want}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="auto", choices=["auto", "cfg2", "cfg3", "cfg4"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-reps", type=int, default=10)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from md_neighbor_list_amd import NeighListGPU, inputs, slab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # Rehearsal on a one-GPU box (NL_BENCH_REHEARSAL=1): all ranks share GPU 0 and talk over gloo (host staging);
    # exercises every line of the N > 1 path except RCCL itself.  Never used by the driver's runs.
    rehearsal = os.environ.get("NL_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    red_dev = "cpu" if rehearsal else dev  # device of the two scalar reductions at the end

    density = 0.5 if args.workload == "cfg3" else 1.0
    if args.workload == "cfg4":
        n_total, scaling = 1 << 25, "strong"
    else:
        n_total, scaling = N_PER_GPU * world, "weak"
    np_dtype, t_dtype = (np.float32, torch.float32) if args.dtype == "f32" else (np.float64, torch.float64)
    if args.workload == "cfg4":
        q, box = inputs.uniform_box(n_total, density, np_dtype)  # the cubic 33.5 M box; every rank generates the same
    else:
        # weak scaling: the single-GPU cube repeated `world` times along z, so that every rank's slab is the N = 1
        # problem plus its two ghost layers (world = 1: the cube itself, BASELINE config 2 / 3)
        q, box = inputs.weak_scaling_box(world, N_PER_GPU, density, np_dtype)
    mesh = slab.mesh_of(box, RC)

    nl = NeighListGPU(RC, *box, dtype=t_dtype, device=dev)
    if world == 1:
        qd = torch.from_numpy(q).to(dev)
        nl.Initialize(n_total)
        step = lambda: nl.MakeNeighList(qd, n_total, sync=False)  # noqa: E731
        st = None
    else:
        st = slab.setup(torch.from_numpy(q).to(dev), None, box, RC, rank, world)
        nl.Initialize(st.q_all.shape[0])
        # the default capacity estimate uses the global density with the local count: size it for the slab
        per = density * (2.0 / 3.0) * np.pi * RC**3
        nl.set_capacity(int(st.n_rows * per * 1.3) + 64 * st.n_rows + 4096)
        step = lambda: slab.build(nl, st, sync=False)  # noqa: E731

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    nl.synchronize()  # surfaces any error (capacity, out of box) before timing
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    fence()
    elapsed = time.perf_counter() - t0
    nl.synchronize()

    npairs_local = nl.half_number_of_pairs()
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    p = torch.tensor([npairs_local], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    npairs = int(p.item())
    ms_per_step = elapsed / args.steps * 1e3

    out = None
    if rank == 0:
        vec_bytes = 16 if args.dtype == "f32" else 32
        # dominant kernel: the list-filling pair search (k_sweep<FILL>); live per-kernel time from HIP events
        # on the launch stream (nl_profile_stages), single-GPU build of this rank's particles
        stages = nl.profile_last_build(reps=args.profile_reps)
        n_loc, p_loc = (n_total if world == 1 else st.n_total), npairs_local
        roofline = None
        if stages:
            # Dominant kernel = the pair search.  With hit masks (default) it is the COUNT_MASKS sweep, which runs
            # every distance test once and decides every list entry; the expansion kernel only places them.  It is
            # charged the algorithmic bytes of the search + append step: sorted positions
            # (16|32 B) + sorted_row (4 B) read, counts (4 B) and the list (4 B per half pair) written.
            info = nl.build_info()
            if info["masks"]:
                kname = ("k_sweep_mfma_f32" if info["mfma"] else "k_sweep_count_masks_f32") if args.dtype == "f32" \
                    else "k_sweep<double,COUNT_MASKS>"
                kms = stages["count"]
            elif stages["fill"] >= stages["count"]:
                kname, kms = "k_sweep<FILL>", stages["fill"]
            else:
                kname, kms = "k_sweep<COUNT>", stages["count"]
            b_search = n_loc * (vec_bytes + 8) + 4 * p_loc
            gbs = b_search / (kms * 1e-3) / 1e9
            copy_gbs = _measured_copy_gbs(torch, dev)
            roofline = {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": round(gbs / copy_gbs, 4),
                        "traffic": _traffic_from_profiles(kname) if world == 1 else None,
                        "algorithmic_bytes_per_launch": b_search, "kernel_ms": round(kms, 4),
                        "note": "VALU-issue bound, not HBM bound: 11.4 VALU + 3.7 SALU per 64 distance tests (DESIGN.md "
                                "section 4); `build` gives the whole-build HBM fraction",
                        "stages_ms": {k: round(v, 4) for k, v in stages.items()}}
        b_build = n_total * vec_bytes + 4 * npairs + 4 * (n_total + 1) + 4 * n_total
        build_gbs = b_build / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "Verlet-list build ms and Mpairs/s at N=1M rho=1.0; achieved HBM GB/s",
            "value": round(npairs / (ms_per_step * 1e-3) / 1e6, 1),
            "unit": "Mpairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            **({"rehearsal": "ranks share one GPU over gloo: not a measurement"} if rehearsal else {}),
            "config": {"workload": f"uniform random box {box[0]:.2f} x {box[1]:.2f} x {box[2]:.2f}, N={n_total}, rho={density}, rc={RC}, "
                                   f"{'fp32 float4' if args.dtype == 'f32' else 'fp64 double4'} positions, half list "
                                   f"(CSR in original particle order), mesh {mesh[0]}x{mesh[1]}x{mesh[2]}",
                       "n_particles": n_total, "half_pairs": npairs,
                       "half_pairs_reference": _reference_pairs(n_total, density, args.dtype, npairs, world, scaling == "weak"),
                       "decomposition": "none" if world == 1 else f"{world} z-slabs + 1-cell ghost layers (p2p)"},
            "build": {"algorithmic_bytes": b_build, "achieved_GBs": round(build_gbs, 1),
                      "frac_of_hbm_peak": round(build_gbs / HBM_PEAK_GBS, 4)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(q, box)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
