#!/usr/bin/env python3
"""bench.py — measurement points per second of the ReMo3D hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the reference's
3D benchmark model — Examples/Benchmark models/Benchmark model 3, dip 30 deg — logged with a
normal and a lateral tool (A0.4M6.0N, A2.0M0.5N) at 100 depths per GPU in [5, 20) m, R = 50 m,
batch_size 5 (=> 40 batches / 200 right-hand sides / 200 measurement points per GPU, computed
with the build's task builder, pinned against the reference's in tests/golden/tasks_bm3.json).
Batch meshes are seeded synthetic half-ball meshes with the reference's size field (no Gmsh in
the image); their sizes (T, n, nnz) are printed in the JSON line.

One "step" = one pass of the hot path over every batch of this rank's share: dof numbering,
CSR pattern, assembly, multi-RHS two-level PCG, axis evaluation, apparent resistivity — then ONE
all-reduce of the log slab across ranks (RCCL).  Mesh arrays, sigma and points are resident on
the device before the timed region (remo_batch_create); mesh generation is excluded, as SURVEY.md
section 8d defines the point.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torchrun, one rank
per GPU.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIZES = {"S": 4.0, "M": 2.5, "L": 1.2, "XL": 0.7}   # multiplier on the reference size field
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: HBM3E 8 TB/s


def build_workload(rank, world, depths_per_gpu, scale, dim=3, mesh_3d="lattice"):
    from remo3d_amd import geometry, tasks
    from remo3d_amd.model import Model, default_mesh_provider
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 3")
    names = ["A0.4M6.0N", "A2.0M0.5N"]
    m = Model(names)
    m.set_model_parameters(os.path.join(ex, "Formation_BM3_30.txt"), os.path.join(ex, "Borehole_BM3.txt"), dip=30)
    m.borehole_model = m._add_points_to_borehole()
    depths = np.linspace(5.0, 20.0, depths_per_gpu * world, endpoint=False)
    sim, batches = tasks.build_batches(m.tools, m.sec, depths, 5)
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    provider = default_mesh_provider(scale=scale, seed=0, mesh_3d=mesh_3d)
    work = []
    t0 = time.time()
    for bi in range(rank, len(batches), world):
        b = batches[bi]
        fg, bh, sigma = geometry.select_data_range(bg, m.formation_model, m.dip_rad, mud[bi], sim[bi], 50.0)
        mesh = provider(dim, 50.0, b, fg, bh, m.dip_rad)
        sources, evals, readers = tasks.batch_rhs(b, m.tools)
        work.append(dict(mesh=mesh, sigma=sigma, sources=sources, evals=evals, readers=readers))
    return dict(model=m, depths=depths, n_batches=len(batches), work=work, mesh_s=time.time() - t0, names=names)


def cpu_baseline(work, rtol, cores=None):
    """The oracle (scalar C port of the same algorithm) the way the reference farms its work out
    (remo3d.py:592-595, worker.py:104-112: one single-threaded worker per core, every right-hand side
    assembled and solved on its own): `cores` host threads, each running assembly + Jacobi-PCG +
    evaluation of ONE right-hand side of the workload at the same time (the C calls release the GIL).
    Reported beside the GPU number, never the target.  Returns (record, potentials of batch 0 / RHS 0)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.fem_oracle import lib, solve_batch
    lib()                                            # compile / load outside the timed span
    if cores is None:
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        cores = max(1, min(avail, 16))
    jobs = []
    for bi, w in enumerate(work):
        for ri in range(len(w["sources"])):
            jobs.append((bi, ri))
    jobs = jobs[:cores]

    def one(job):
        w = work[job[0]]
        z, I = w["sources"][job[1]]
        ez = list(w["evals"][job[1]])
        out, rc, st = solve_batch(w["mesh"], w["sigma"], [0, len(z)], list(z), list(I), [0, len(ez)], ez, condense=True, rtol=rtol, maxit=1000)
        return out, st["iterations"]

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        res = list(ex.map(one, jobs))
    dt = time.time() - t0
    pts = sum(len(work[bi]["readers"][ri]) for bi, ri in jobs)
    its = [r[1] for r in res]
    return dict(value=pts / dt, unit="points/s", cores=len(jobs), kind="port",
                sample=f"the first {len(jobs)} right-hand sides of the workload ({pts} points), one per host thread at the same time, "
                       f"each its own oracle/fem_oracle.c assembly + Jacobi-PCG (rtol {rtol:g}, {min(its)}-{max(its)} steps) + evaluation: "
                       f"{dt:.1f} s wall; NGSolve is not installable here, so this is the build's scalar C restatement run one worker per "
                       "core like the reference's farm, not the reference binary"), res[0][0]


def pmc_traffic(workload, n_free, nnz):
    """HBM bytes per SpMM launch from the committed rocprofv3 --pmc passes of this exact workload
    (profiles/, collected with tools/collect_traffic.sh + tools/pmc_traffic.py); None when the run differs."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_f_pmc_traffic_default_bench.json")) as f:
            p = json.load(f)
    except OSError:
        return None
    if p.get("workload") == workload and p.get("n_free") == n_free and p.get("nnz") == nnz:
        return p["spmm"]["traffic_bytes_per_launch"]
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", default="S", choices=list(SIZES))
    ap.add_argument("--depths", type=int, default=100, help="measurement depths per GPU")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--maxsteps", type=int, default=1000)
    ap.add_argument("--mesh", default="lattice", choices=["lattice", "conforming"],
                    help="lattice = the seeded synthetic half-ball meshes of SURVEY 8d (headline workload); conforming = the interface-"
                         "conforming revolved meshes Model uses by default for dipping models")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "mixed"],
                    help="fp64 (the headline configuration) or mixed = fp32 PCG inside fp64 refinement (BASELINE config 5)")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts (HIP streams + arenas) per GPU, each driven by its own host thread over its share of the batches; "
                         "1 = the headline configuration (per-launch SpMM timing is only meaningful without overlap)")
    ap.add_argument("--overlap", default="prepare", choices=["prepare", "all"],
                    help="with --streams > 1: 'prepare' = only one batch is in its PCG at a time, the other contexts number / assemble "
                         "theirs beside it; 'all' = no restriction")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B experiments only: remo_debug_tune(KEY, VALUE) before the run (include/remo3d_hip.h lists the keys)")
    ap.add_argument("--no-events", action="store_true", help="do not bracket SpMV launches with HIP events")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket every k-th SpMV launch of a solve with HIP events (a bracket costs the stream ~1.5 us: bracketing "
                         "every launch takes 6 %% off the throughput it is there to explain)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("REMO_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # REMO_DEVICE: rehearsals on a 1-GPU box
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    from remo3d_amd import solver, sweep, tasks
    if args.tune:
        from remo3d_amd import _lib
        for kv in args.tune:
            key, value = kv.split("=")
            _lib.load().remo_debug_tune(int(key), int(value))
    dist_on = False
    if world > 1:
        import torch
        dist_on = sweep.init_from_env()
        if torch.cuda.is_available():
            torch.cuda.set_device(local)

    wl = build_workload(rank, world, args.depths, SIZES[args.size], mesh_3d=args.mesh)
    work = wl["work"]
    ctx = solver.Context(local)
    opts = solver.make_opts(preconditioner="multigrid", condense=True, rtol=args.rtol, maxsteps=args.maxsteps,
                            time_kernels=0 if args.no_events else max(1, args.event_stride), precision=args.precision,
                            serialize_solves=(args.streams > 1 and args.overlap == "prepare"))
    ctxs = [ctx] + [solver.Context(local) for _ in range(max(1, args.streams) - 1)]
    resident = [ctxs[i % len(ctxs)].batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for i, w in enumerate(work)]
    n_tools = len(wl["names"])
    pool = None
    if len(ctxs) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=len(ctxs))

    def one_step():
        slab = np.zeros((len(wl["depths"]), n_tools))
        agg = dict(spmv_ms=0.0, spmv_launches=0, spmv_bytes_total=0.0, pcg_steps=0, not_converged=0, ms_symbolic=0.0, ms_assemble=0.0,
                   ms_solve=0.0, ms_h2d=0.0, ms_eval=0.0, n=0, nnz=0, max_it=0)
        if pool is not None:   # one host thread per context, each walks its own batches in order (ctypes releases the GIL)
            def drive(j):
                return [(i, resident[i].run(opts, raise_on_error=False)) for i in range(j, len(resident), len(ctxs))]
            rcs = dict(p for chunk in pool.map(drive, range(len(ctxs))) for p in chunk)
        for i, (w, b) in enumerate(zip(work, resident)):
            rc = rcs[i] if pool is not None else b.run(opts, raise_on_error=False)
            st = b.stats
            if rc < 0:
                for rd in w["readers"]:
                    for (di, ti, K, o, m) in rd:
                        slab[di, ti] = np.nan
                continue
            agg["not_converged"] += int(rc == 1)
            outs = b.fetch()
            for u, rd in zip(outs, w["readers"]):
                for (di, ti, K, o, m) in rd:
                    slab[di, ti] = tasks.apparent_resistivity(u[o:o + m], m, K, 3)
            agg["spmv_ms"] += st["spmv_ms"]; agg["spmv_launches"] += st["spmv_launches"]
            agg["spmv_ms_raw"] = agg.get("spmv_ms_raw", 0.0) + st["spmv_ms_raw"]; agg["ev_over"] = st["event_overhead_ms"]
            agg["spmv_bytes_total"] += st["spmv_bytes"] * st["spmv_launches"]
            agg["pcg_steps"] += st["pcg_steps"]; agg["max_it"] = max(agg["max_it"], st["max_iterations"])
            for k in ("ms_symbolic", "ms_assemble", "ms_solve", "ms_h2d", "ms_eval"):
                agg[k] += st[k]
            agg["n"] = st["n_free"]; agg["nnz"] = st["nnz"]
        slab = sweep.combine(slab)   # the ONE collective of the path: all-reduce of the log slab
        return slab, agg

    def sync():
        if dist_on:
            import torch
            sweep.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.time()
    for _ in range(args.steps):
        slab, agg = one_step()
    sync()
    dt = sweep.max_over_ranks(time.time() - t0)

    n_points = len(wl["depths"]) * n_tools
    value = n_points * args.steps / dt
    workload_name = f"BM3 dip30, tools A0.4M6.0N+A2.0M0.5N, {args.depths} depths/GPU, R=50, batch 5, mesh size {args.size}"
    if args.mesh != "lattice":
        workload_name += ", interface-conforming revolved meshes"
    if rank != 0:
        return
    ach = (agg["spmv_bytes_total"] / 1e9) / (agg["spmv_ms"] / 1e3) if agg["spmv_ms"] > 0 else None
    roofline = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=(ach / HBM_PEAK_GBS) if ach else None,
                    traffic=pmc_traffic(workload_name,
                                        int(agg["n"]), int(agg["nnz"])) if args.precision == "fp64" else None,
                    kernel="k_spmm_pair (CSR SpMM, %s, k=5 interleaved RHS)" % ("fp64" if args.precision == "fp64" else "fp32 values and vectors"),
                    timed="every %d-th launch of every solve, HIP events on the solver's stream, over the timed steps" % max(1, args.event_stride), launches=int(agg["spmv_launches"]),
                    avg_launch_us=(1e3 * agg["spmv_ms"] / agg["spmv_launches"]) if agg["spmv_launches"] else None,
                    avg_bracket_us_raw=(1e3 * agg.get("spmv_ms_raw", 0.0) / agg["spmv_launches"]) if agg["spmv_launches"] else None,
                    empty_event_pair_us=1e3 * agg.get("ev_over", 0.0),
                    bytes_per_launch="12*nnz + 4*n + 16*k*n (SURVEY.md 8d)" if args.precision == "fp64" else "8*nnz + 4*n + 8*k*n (SURVEY.md 8d, fp32 storage)", traffic_unit="bytes per launch: reads sized by the TCC_EA0_RDREQ 32/64/128-B request counters + WRITE_SIZE, profiles/r01_f_pmc_traffic_default_bench.json")
    out = dict(metric="measurement points/sec (3D benchmark model)", value=value, unit="points/s", n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f64" if args.precision == "fp64" else "f32 PCG inside f64 residual refinement", data="synthetic",
               config=dict(workload=workload_name,
                           batches_per_gpu=len(work), rhs_per_gpu=sum(len(w["sources"]) for w in work), points_total=n_points,
                           mesh_T=int(work[0]["mesh"].n_elems), n_free=int(agg["n"]), nnz=int(agg["nnz"]), rtol=args.rtol,
                           maxsteps=args.maxsteps, precision=args.precision, streams_per_gpu=len(ctxs), preconditioner="multigrid = Chebyshev polynomial on the P1 vertex block (degree / interval by vertex count: 5 on lmax/90..lmax at 12.6 k vertices, 13 on lmax/320 at 83 k) + Jacobi on edge/face dofs", max_pcg_iterations=int(agg["max_it"]),
                           batches_not_converged=int(agg["not_converged"]), nan_points=int(np.isnan(slab).sum())),
               roofline=roofline,
               breakdown_ms_per_step=dict(symbolic_host=agg["ms_symbolic"], h2d=agg["ms_h2d"], assemble=agg["ms_assemble"], solve=agg["ms_solve"],
                                          eval=agg["ms_eval"], pcg_steps=int(agg["pcg_steps"]), mesh_generation_excluded_s=wl["mesh_s"]))
    if args.tune:
        out["config"]["debug_tune"] = list(args.tune)
    if not args.no_cpu and world == 1:           # the CPU leg belongs to the N = 1 line only
        cb, ref_out = cpu_baseline(work, args.rtol)
        out["cpu_baseline"] = cb
        got = resident[0].fetch()[0]
        out["config"]["gpu_vs_oracle_max_rel_diff_batch0_rhs0"] = float(np.max(np.abs(got - ref_out) / np.abs(ref_out)))
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
