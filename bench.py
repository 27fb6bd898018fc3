#!/usr/bin/env python3
"""Benchmark of the Verlet-list build (BASELINE.json metric), one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one neighbour-list build over positions already resident in HBM (the reference's timing loop,
make_list.cu:124-127: MakeNeighList(q, N, sync=false) back to back, one device sync at the end).

  N = 1   workload = BASELINE config 2: 1 048 576 particles, rho = 1.0, rc = 3.3, fp32, uniform random box
          (``--workload cfg3 | cfg4 | cfg5`` select the other single-device configurations: rho = 0.5; the
          33 554 432-particle box on ONE device with 64-bit list offsets; fp64 at 2 x cut-off).
  N > 1   workload = BASELINE config 4, STRONG scaling: the 33 554 432-particle box cut into N z-slabs of whole cell
          layers, one rank per GPU; every step does the ghost-layer exchange (point-to-point over RCCL/xGMI) and the
          slab build.  ``--workload weak`` instead gives every rank the 1 M-particle cube (N x 1 048 576 particles).

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      the WHOLE build against the HBM roofline: algorithmic bytes B = N*sizeof(Vec) + 4P + 4(N+1) + 4N
                 over ms_per_step; `kernels` = every stage with the bytes that stage itself moves
                and its HIP-event time; `valu` = the vector-issue ceiling of the dominant kernel (the path sits on the
                compute side of the ridge: both ceilings are reported)
  cpu_baseline  the reference's own CPU classes (oracle/_ref, compiled from the reference), 1 pinned core, same input
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS, CLK_HZ, VALU_CYCLES = 256 * 4, 2.4e9, 2.0  # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles
N_PER_GPU = 1 << 20


def cpu_baseline(q32, box, rc, budget_s=90.0):
    """Times the reference CPU path on this host (rank 0, N = 1 only), BASELINE.md section 2: one pinned core
    (sched_setaffinity, the `taskset -c` of the survey), 1 warm-up build + 3 timed builds of the same 1 M workload per
    reference class.  kind 'reference' = the reference's classes compiled from its sources (oracle/_ref); 'port' = the C
    restatement when those are absent."""
    from oracle import pyoracle as po

    n = len(q32)
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        core = sorted(old_aff)[0]
        os.sched_setaffinity(0, {core})
    except (AttributeError, OSError):
        core = None
    try:
        variants = {}
        t_start = time.time()
        q64 = None
        for name, label in (("avx512_8x1", "NeighListAVX512 8x1 (fp64 only)"),
                            ("fused_native", "scalar NeighList<float3> -DLOOP_FUSION (fp32)"),
                            ("avx2_4x1", "NeighListAVX2 4x1 (fp64 only)")):
            if not po.ref_runnable(name) or time.time() - t_start > budget_s:
                continue
            if name == "fused_native":
                qq = q32
            else:
                if q64 is None:
                    q64 = q32.astype(np.float64)
                qq = q64
            _, warm, npairs = po.ref_build(qq, rc, box, name, loops=1, want_list=False)
            timed = 3 if warm < 8.0 else 1  # a slow host (this container: 4-13 s per build) keeps to one timed build
            _, secs, npairs = po.ref_build(qq, rc, box, name, loops=timed, want_list=False)
            secs /= timed
            variants[name] = {"what": label, "seconds_per_build": round(secs, 3), "warmup_builds": 1, "timed_builds": timed,
                              "npairs": npairs, "mpairs_per_s": round(npairs / secs / 1e6, 3)}
        if variants:
            best = max(variants.values(), key=lambda v: v["mpairs_per_s"])
            return {"value": best["mpairs_per_s"], "unit": "Mpairs/s", "cores": 1, "kind": "reference",
                    "pinned_to_core": core,
                    "sample": f"1 warm-up + up to 3 timed builds of the full workload (N={n}, rc={rc}) per reference class, "
                              f"one pinned thread; value = fastest class ({best['what']})",
                    "variants": variants, "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}
        t0 = time.time()
        h = po.build(q32, rc, box)
        secs = time.time() - t0
        return {"value": round(h.npairs / secs / 1e6, 3), "unit": "Mpairs/s", "cores": 1, "kind": "port",
                "sample": f"1 build of the full workload (N={n}) with the C restatement oracle/nl_oracle.c",
                "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}
    finally:
        if old_aff is not None:
            try:
                os.sched_setaffinity(0, old_aff)
            except OSError:
                pass


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


def _newest_pmc(workload):
    """The newest committed PMC passes for this workload (profiles/rNN_vMM[_<workload>]_pmc.json; round, then version,
    compared as NUMBERS), or None.  cfg2 files carry no workload suffix."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    if os.path.isdir(pdir):
        for f in os.listdir(pdir):
            m = re.fullmatch(r"r(\d+)_v(\d+)(?:_(cfg\d))?_pmc\.json", f)
            if not m or (m.group(3) or "cfg2") != workload:
                continue
            key = (int(m.group(1)), int(m.group(2)))
            if best is None or key > best[0]:
                best = (key, os.path.join(pdir, f))
    if not best:
        return None, None
    try:
        return json.load(open(best[1])), os.path.basename(best[1])
    except Exception:
        return None, None


def _reference_pairs(key, got):
    """The pair count of this box from tests/golden/known_answers.json (compiled reference, or the restatement's count
    mode where the reference overflows; a committed fixture, nothing under oracle/ runs here): "ok" when the build (the
    union over ranks for N > 1) found exactly that many pairs, the two numbers otherwise, None when no answer is stored."""
    try:
        ka = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    except Exception:
        return None
    if key not in ka:
        return None
    want = int(ka[key]["npairs"])
    return "ok" if want == int(got) else {"got": int(got), "reference": want}


# name -> (density, rc, dtype, particles, known-answer key)
WORKLOADS = {
    "cfg2": (1.0, 3.3, "f32", 1 << 20, "u1M_rho1_f32"),
    "cfg3": (0.5, 3.3, "f32", 1 << 20, "u1M_rho05_f32"),
    "cfg4": (1.0, 3.3, "f32", 1 << 25, "u32M_rho1_f32"),
    "cfg5": (1.0, 6.6, "f64", 1 << 20, "u1M_rho1_f64_rc66"),
}


def _stage_table(stages, info, n, p, vec_bytes, pos_bytes, full=False):
    """Every stage of the build with the bytes THAT stage moves by design (reads + writes of its own arrays; cache
    re-reads of the stencil are not counted) and its HIP-event time.  n = particles on this device, p = list entries."""
    rows = []
    nb = max(1, info.get("mask_rows", 1))
    masks = 192 * n * nb if info["masks"] else 0

    def add(name, key, nbytes, what):
        ms = stages.get(key, 0.0)
        if ms <= 0:
            return
        gbs = nbytes / (ms * 1e-3) / 1e9
        rows.append({"stage": name, "ms": round(ms, 4), "bytes_moved": int(nbytes), "what": what,
                     "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)})

    add("binning 1 (k_bin_rows)", "hash", vec_bytes * n, "positions read")
    add("binning 2+3 (k_bin_scatter, k_bin_cells)", "reorder", vec_bytes * n + 2 * (pos_bytes + 4) * n + (pos_bytes + 8) * n,
        "positions read; row-grouped copy written and read; sorted positions + row + id written")
    add("pair search COUNT" + (" keeping hit masks" if info["masks"] else ""), "count", (pos_bytes + 4) * n + 4 * n + masks,
        "sorted positions + rows read once, counts" + (f" and up to {192 * nb} B of hit masks per particle (128 B where the stream has at most 16 tiles)" if info["masks"] else "") + " written")
    add("row scan (k_scan_chained)", "row_scan", 4 * n + 4 * n, "counts read, key_pointer written")
    if info["masks"]:
        if nb > 1:
            add("expansion (k_row_base, k_fill_dense)", "fill", masks + 12 * n + 4 * n + 4 * p,
                "hit masks, ids, row offsets read; the list written")
        else:
            add("expansion (" + ("k_fill_rows" if info.get("fine_rows") else "k_fill_masks") + ")", "fill", masks + 12 * n + 4 * p,
                "hit masks, ids, row ids and their list offsets read; the list written")
    else:
        add("pair search FILL", "fill", (pos_bytes + 4) * n + 4 * n + 4 * p, "sorted positions + offsets read; the list written")
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batches", type=int, default=7,
                    help="timed batches of --steps builds each (SURVEY.md section 8d: the reference times LOOP = 100 builds, make_list.cu:20,124-132); "
                         "ms_per_step is the median batch, min and max are reported next to it")
    ap.add_argument("--workload", default="auto", choices=["auto", "cfg2", "cfg3", "cfg4", "cfg5", "weak"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cfg4-baseline", action="store_true",
                    help="N = 1: skip the extra measurement of the 33.5 M box on one device (the N = 1 point of the strong-scaling series)")
    ap.add_argument("--profile-reps", type=int, default=10)
    ap.add_argument("--dist", default="cabi", choices=["cabi", "torch"],
                    help="N > 1: the whole decomposed build inside the library behind the C ABI (default: nl_make_list_distributed -- pack "
                         "kernel, fixed-capacity halo messages over RCCL, unpack, slab build; no host synchronisation per build), or the halo "
                         "exchange by torch.distributed p2p around nl_make_list_slab (md_neighbor_list_amd/slab.py)")
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
        # A watchdog for the multi-rank run: the N > 1 exchange over RCCL has never run on hardware;
        # should a rank ever wait for a message that does not come, the job ends with an error after NL_BENCH_WATCHDOG_S
        # seconds (default 600) instead of holding the node until the driver's limit.
        import signal

        def _watchdog(signum, frame):
            sys.stderr.write(f"[rank {rank}] bench.py watchdog: no result after the time limit -- a rank is stuck in the halo exchange\n")
            sys.stderr.flush()
            os._exit(124)

        signal.signal(signal.SIGALRM, _watchdog)
        signal.alarm(int(os.environ.get("NL_BENCH_WATCHDOG_S", "600")))
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    red_dev = "cpu" if rehearsal else dev  # device of the scalar reductions at the end

    wl = args.workload
    if wl == "auto":
        wl = "cfg2" if world == 1 else "cfg4"
    if wl == "weak":
        density, rc, dtype, ka_key = 1.0, 3.3, "f32", ("u1M_rho1_f32" if world == 1 else f"w{world}x1M_rho1_f32")
        n_total, scaling = N_PER_GPU * world, "weak"
    else:
        density, rc, dtype, n_total, ka_key = WORKLOADS[wl]
        scaling = "strong"
    if os.environ.get("NL_BENCH_N"):  # rehearsals on small boxes only (never the driver's runs): total particle count
        n_total, ka_key = int(os.environ["NL_BENCH_N"]), None
    np_dtype, t_dtype = (np.float32, torch.float32) if dtype == "f32" else (np.float64, torch.float64)
    if wl == "weak":
        # the single-GPU cube repeated `world` times along z: every rank's slab is the N = 1 problem plus two ghost layers
        q, box = inputs.weak_scaling_box(world, N_PER_GPU, density, np_dtype)
    else:
        q, box = inputs.uniform_box(n_total, density, np_dtype)  # every rank generates the same box
    mesh = slab.mesh_of(box, rc)

    nl = NeighListGPU(rc, *box, dtype=t_dtype, device=dev)
    if world == 1:
        qd = torch.from_numpy(q).to(dev)
        nl.Initialize(n_total)
        step = lambda: nl.MakeNeighList(qd, n_total, sync=False)  # noqa: E731
        st = None
    else:
        dn = None
        per = density * (2.0 / 3.0) * np.pi * rc**3
        if args.dist == "cabi":
            from md_neighbor_list_amd.dist import DistributedNeighList

            # If the library's communicator cannot be set up or its first (synchronous) build fails on every rank alike
            # -- RCCL not loadable, an API error -- all ranks fall back to the torch.distributed exchange together and
            # the line says so.  (An error on one rank only would leave the others inside RCCL: nothing to catch there.)
            ok = 1
            try:
                dn = DistributedNeighList(nl, rank, world, transport="host" if rehearsal else "rccl")
                nl.Initialize(int(n_total / world * 1.6) + 65536)
                n_own = dn.scatter(torch.from_numpy(q).to(dev), box, rc)
                nl.set_capacity(int(n_own * per * 1.3) + 64 * n_own + 4096)
                dn.build(sync=True)
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] nl_make_list_distributed unavailable ({e}); falling back to --dist torch", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                args.dist, dn = "torch(fallback)", None
                nl = NeighListGPU(rc, *box, dtype=t_dtype, device=dev)
        if dn is not None:
            step = lambda: dn.build(sync=False)  # noqa: E731
            st = None
        else:
            st = slab.setup(torch.from_numpy(q).to(dev), None, box, rc, rank, world)
            nl.Initialize(st.q_all.shape[0])
            # the default capacity estimate uses the global density with the local count: size it for the slab
            nl.set_capacity(int(st.n_rows * per * 1.3) + 64 * st.n_rows + 4096)
            step = lambda: slab.build(nl, st, sync=False)  # noqa: E731

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    nl.synchronize()  # surfaces any error (capacity, out of box) before timing
    # every batch: exactly --steps builds between a barrier + device synchronisation on both sides
    batch_s = []
    for _ in range(max(1, args.batches)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        fence()
        batch_s.append(time.perf_counter() - t0)
    nl.synchronize()

    npairs_local = nl.half_number_of_pairs()
    cs_local, _ = nl.list_checksum()
    t = torch.tensor(batch_s, dtype=torch.float64, device=red_dev)
    p = torch.tensor([npairs_local, cs_local >> 32, cs_local & 0xFFFFFFFF], dtype=torch.int64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # a batch takes as long as its slowest rank
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
    batch_ms = sorted(float(x) / args.steps * 1e3 for x in t.tolist())
    npairs = int(p[0].item())
    checksum = ((int(p[1].item()) << 32) + int(p[2].item())) & ((1 << 64) - 1)  # wrapping sum of the ranks' checksums
    ms_per_step = batch_ms[len(batch_ms) // 2]  # the median batch

    out = None
    if rank == 0:
        vec_bytes = 16 if dtype == "f32" else 32
        pos_bytes = vec_bytes
        b_build = n_total * vec_bytes + 4 * npairs + 4 * (n_total + 1) + 4 * n_total
        build_gbs = b_build / (ms_per_step * 1e-3) / 1e9
        # per-stage device time from HIP events on the launch stream (nl_profile_last_build), this rank's build
        stages = nl.profile_last_build(reps=args.profile_reps)
        info = nl.build_info()
        n_loc, p_loc = (n_total if world == 1 else st.n_total if st is not None else nl._n), npairs_local
        kernels = _stage_table(stages, info, n_loc, p_loc, vec_bytes, pos_bytes) if stages else []
        pmc, pmc_file = _newest_pmc(wl) if world == 1 else (None, None)
        traffic, valu = None, None
        dom = max(kernels, key=lambda k: k["ms"]) if kernels else None
        if pmc:
            per_launch = pmc.get("hbm_bytes_per_launch", {})
            traffic = float(sum(per_launch.values())) if per_launch else None
            # vector-issue ceiling of the dominant kernel: SQ_INSTS_VALU wave-instructions x 2 cycles over SIMDs x time x clock
            cnt = pmc.get("counters", {})
            dk = max(cnt, key=lambda k: cnt[k].get("avg_us", 0.0)) if cnt else None
            if dk and cnt[dk].get("SQ_INSTS_VALU") and cnt[dk].get("avg_us"):
                insts, us = float(cnt[dk]["SQ_INSTS_VALU"]), float(cnt[dk]["avg_us"])
                valu = {"kernel": dk, "insts": int(insts), "salu_insts": int(cnt[dk].get("SQ_INSTS_SALU", 0)), "kernel_us": round(us, 1),
                        "frac_of_issue_peak": round(insts * VALU_CYCLES / (SIMDS * us * 1e-6 * CLK_HZ), 4),
                        "model": f"SQ_INSTS_VALU x {VALU_CYCLES:.0f} cycles / ({SIMDS} SIMDs x t x {CLK_HZ / 1e9:.1f} GHz), from {pmc_file}"}
        copy_gbs = _measured_copy_gbs(torch, dev)
        roofline = {"bound": "hbm", "scope": "whole build (every kernel of one MakeNeighList)",
                    "achieved": round(build_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(build_gbs / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes": b_build,
                    "traffic": traffic, "traffic_source": pmc_file,
                    "measured_copy_GBs": copy_gbs, "frac_of_measured_copy": round(build_gbs / copy_gbs, 4),
                    "dominant_kernel": dom["stage"] if dom else None,
                    "kernels": kernels, "valu": valu,
                    "stages_ms": {k: round(v, 4) for k, v in stages.items()} if stages else None}
        out = {
            "metric": "Verlet-list build ms and Mpairs/s at N=1M rho=1.0; achieved HBM GB/s",
            "value": round(npairs / (ms_per_step * 1e-3) / 1e6, 1),
            "unit": "Mpairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_min": round(batch_ms[0], 4), "ms_per_step_max": round(batch_ms[-1], 4), "batches": len(batch_ms),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            **({"rehearsal": "ranks share one GPU over gloo: not a measurement"} if rehearsal else {}),
            "config": {"workload": f"{wl}: uniform random box {box[0]:.2f} x {box[1]:.2f} x {box[2]:.2f}, N={n_total}, rho={density}, rc={rc}, "
                                   f"{'fp32 float4' if dtype == 'f32' else 'fp64 double4'} positions, half list "
                                   f"(CSR in original particle order), mesh {mesh[0]}x{mesh[1]}x{mesh[2]}",
                       "n_particles": n_total, "half_pairs": npairs, "list_checksum": f"{checksum:016x}",
                       "half_pairs_reference": _reference_pairs(ka_key, npairs) if ka_key else None,
                       "offset_bits": info["offset_bits"],
                       "decomposition": "none" if world == 1 else f"{world} z-slabs + 1-cell ghost layers (p2p; "
                                        + ("nl_make_list_distributed" if args.dist == "cabi" else "torch.distributed around nl_make_list_slab"
                                           + (" -- FALLBACK: the library's communicator failed" if "fallback" in args.dist else "")) + ")"},
            "roofline": roofline,
        }
        if ka_key:
            try:
                want = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))[ka_key]["hash"]
                out["config"]["list_checksum_reference"] = "ok" if want == f"{checksum:016x}" else {"reference": want}
            except Exception:
                pass
    if world == 1 and rank == 0 and wl == "cfg2" and not args.no_cfg4_baseline and not os.environ.get("NL_BENCH_N"):
        # The N = 1 point of the strong-scaling series (N > 1 runs BASELINE config 4): the same 33.5 M box on ONE device.
        try:
            del nl, qd
            torch.cuda.empty_cache()
            d4, rc4, _, n4, key4 = WORKLOADS["cfg4"]
            q4, box4 = inputs.uniform_box(n4, d4, np.float32)
            nl4 = NeighListGPU(rc4, *box4, dtype=torch.float32, device=dev)
            qd4 = torch.from_numpy(q4).to(dev)
            nl4.Initialize(n4)
            for _ in range(2):
                nl4.MakeNeighList(qd4, n4, sync=False)
            nl4.synchronize()
            torch.cuda.synchronize()
            k4 = 5
            t4 = time.perf_counter()
            for _ in range(k4):
                nl4.MakeNeighList(qd4, n4, sync=False)
            torch.cuda.synchronize()
            ms4 = (time.perf_counter() - t4) / k4 * 1e3
            p4 = nl4.half_number_of_pairs()
            b4 = n4 * 16 + 4 * p4 + 4 * (n4 + 1) + 4 * n4
            out["cfg4_single_gpu"] = {"what": "BASELINE config 4 (N=33554432, rho=1.0, rc=3.3, fp32) on ONE device: the N=1 point of "
                                              "the strong-scaling series that --gpus N>1 measures", "steps": k4,
                                      "ms_per_step": round(ms4, 3), "value": round(p4 / (ms4 * 1e-3) / 1e6, 1), "unit": "Mpairs/s",
                                      "half_pairs": p4, "half_pairs_reference": _reference_pairs(key4, p4),
                                      "offset_bits": nl4.build_info()["offset_bits"],
                                      "frac_of_hbm_peak": round(b4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            del nl4, qd4, q4
            torch.cuda.empty_cache()
        except Exception as e:  # never lose the headline line over the extra measurement
            out["cfg4_single_gpu"] = {"error": repr(e)[:200]}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            if wl == "cfg5":
                # the reference's classes overrun their MAX_PARTNERS*N buffers at 2 x cut-off (neighlist_cpu.hpp:37,76-78):
                # the restatement's count mode (same cells, same visits, same test; OpenMP over cells) stands in
                from oracle import pyoracle as po

                t0 = time.time()
                _, _, _, np5 = po.count(q, rc, box)
                secs = time.time() - t0
                out["cpu_baseline"] = {"value": round(np5 / secs / 1e6, 3), "unit": "Mpairs/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"1 pass of the full workload (N={n_total}, rc={rc}, fp64) with the C restatement in "
                                                 "count mode (oracle/nl_oracle_impl.h nl_oracle_count), all host threads",
                                       "host_cpu": _cpu_model(), "host_cores": os.cpu_count()}
            else:
                out["cpu_baseline"] = cpu_baseline(q, box, rc)
        print(json.dumps(out), flush=True)
    if world > 1:
        signal.alarm(0)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
