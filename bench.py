#!/usr/bin/env python3
"""bench.py — measurement points per second of the ReMo3D hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the reference's
3D benchmark model — Examples/Benchmark models/Benchmark model 3, dip 30 deg — logged with a
normal and a lateral tool (A0.4M6.0N, A2.0M0.5N) at 100 depths per GPU in [5, 20) m, R = 50 m,
batch_size 5 (=> 40 batches / 200 right-hand sides / 200 measurement points per GPU, computed
with the build's task builder, pinned against the reference's in tests/golden/tasks_bm3.json).
Batch meshes are seeded synthetic half-ball meshes with the reference's size field (no Gmsh in
the image), size L = 1.2 x that field; their sizes (T, n, nnz) are printed in the JSON line.

One "step" = one pass of the hot path over every batch of the sweep, the way the product runs it
(`Model` / worker.py:74-142): per batch host arrays -> device (remo_solve_batch), dof numbering, the
assembly the operator needs, multi-RHS two-level PCG, axis evaluation, potentials -> host, apparent
resistivity — FIVE contexts per GPU (streams + arenas, one host thread each; eight hardware queues: remo3d_amd/__init__.py) draw the
batches one after the other, as `Model` does by default — then ONE all-reduce of the log slab across ranks (RCCL).  The timed span is
SURVEY.md 8d's: H2D of the mesh arrays ... D2H of the potentials; mesh generation is excluded (8d) and
reported beside it.  `value` comes from that span.  Per-kernel figures (`roofline`, `breakdown`) come
from a second, single-context leg with the batches resident (kernels that share the chip cannot be
timed in isolation); the line also carries that leg's points/s, further mesh sizes and the
interface-conforming meshes `Model` builds (`sizes`), `Model.compute_synthetic_logs` end to end with
meshing included (`model_end_to_end`), and the CPU oracle on the host cores (`cpu_baseline`).

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 either launched by torchrun (one
rank per GPU), or started plainly: the parent then spawns the N ranks itself (torch.distributed.run)
before it has touched torch or the GPU, relays rank 0's JSON line and the children's exit status.
`--gpus N` never runs on fewer than N ranks.  Rank 0 prints ONE JSON line.

Scaling modes: default weak (`--depths` per GPU, BASELINE configs[2] per GPU); `--total-depths D` = strong
(BASELINE configs[3]: 1000 depths over the ranks).  `--schedule dynamic`: ranks draw batches from a shared
counter (the reference's pull scheduling, remo3d.py:843-860) instead of block-cyclic shares.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIZES = {"S": 4.0, "M": 2.5, "L": 1.2, "XL": 0.7}   # multiplier on the reference size field
HBM_PEAK_GBS = 8000.0                                # MI355X_MICROARCH.md: HBM3E 8 TB/s
TOOLS = ["A0.4M6.0N", "A2.0M0.5N"]
PMC_FILE = os.path.join("profiles", "r04_pmc_traffic_default_bench.json")


def _model_and_batches(n_depths):
    import numpy as np
    from remo3d_amd import tasks
    from remo3d_amd.model import Model
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 3")
    m = Model(TOOLS)
    m.set_model_parameters(os.path.join(ex, "Formation_BM3_30.txt"), os.path.join(ex, "Borehole_BM3.txt"), dip=30)
    m.borehole_model = m._add_points_to_borehole()
    depths = np.linspace(5.0, 20.0, n_depths, endpoint=False)
    sim, batches = tasks.build_batches(m.tools, m.sec, depths, 5)
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    return m, depths, sim, batches, mud, bg


def _build_some(job):
    """Meshes + right-hand sides of the batches `indices` (runs in the caller or in a spawned CPU-only process)."""
    n_depths, scale, mesh_3d, indices = job
    from remo3d_amd import geometry, tasks
    from remo3d_amd.model import default_mesh_provider
    m, depths, sim, batches, mud, bg = _model_and_batches(n_depths)
    provider = default_mesh_provider(scale=scale, seed=0, mesh_3d=mesh_3d)
    out = []
    trace = os.environ.get("REMO_BENCH_TRACE_MESH") == "1"
    for bi in indices:
        b = batches[bi]
        fg, bh, sigma = geometry.select_data_range(bg, m.formation_model, m.dip_rad, mud[bi], sim[bi], 50.0)
        t0 = time.time()
        if trace:
            import faulthandler
            faulthandler.dump_traceback_later(60, repeat=False)
            sys.stderr.write("[mesh worker %d] start scale %.2f %s batch %d\n" % (os.getpid(), scale, mesh_3d, bi)); sys.stderr.flush()
        mesh = provider(3, 50.0, b, fg, bh, m.dip_rad)
        if trace:
            faulthandler.cancel_dump_traceback_later()
            sys.stderr.write("[mesh worker %d] done  scale %.2f %s batch %d: T = %d in %.1f s\n" % (os.getpid(), scale, mesh_3d, bi, mesh.n_elems, time.time() - t0)); sys.stderr.flush()
        sources, evals, readers = tasks.batch_rhs(b, m.tools)
        out.append(dict(index=bi, mesh=mesh, sigma=sigma, sources=sources, evals=evals, readers=readers))
    return out


def _model_and_batches_2d():
    import numpy as np
    from remo3d_amd import tasks
    from remo3d_amd.model import Model
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 1")
    m = Model(["A0.4M6.0N"])
    m.set_model_parameters(os.path.join(ex, "Formation_BM1.txt"), os.path.join(ex, "Borehole_BM1.txt"))
    depths = np.linspace(5, 55, 100)
    sim, batches = tasks.build_batches(m.tools, m.sec, depths, 5)
    mud = np.interp(sim, m.borehole_model[:, 0], m.borehole_model[:, 2])
    bg = np.ascontiguousarray(m.borehole_model[:, :2])
    return m, depths, sim, batches, mud, bg


def _build_some_2d(indices):
    from remo3d_amd import geometry, tasks
    from remo3d_amd.model import default_mesh_provider
    m, depths, sim, batches, mud, bg = _model_and_batches_2d()
    provider = default_mesh_provider()
    out = []
    for bi in indices:
        fg, bh, sigma = geometry.select_netgen_data_range(bg, m.formation_model, mud[bi], sim[bi], 50.0)
        sources, evals, readers = tasks.batch_rhs(batches[bi], m.tools)
        out.append(dict(index=bi, mesh=provider(2, 50.0, batches[bi], fg, bh, 0.0), sigma=sigma, sources=sources, evals=evals, readers=readers))
    return out


class Pending:
    """A workload whose meshes are being built in the pool: one job per batch, all workloads of the run submitted before any
    is waited for (the pool stays full; a two-batch XL workload no longer holds two workers while eight idle)."""

    def __init__(self, meta, futs, t0):
        self.meta, self.futs, self.t0 = meta, futs, t0

    def result(self, note=None):
        work = []
        for f in self.futs:
            work.extend(f.result())
            if note:
                note(len(work), len(self.futs))
        self.meta["work"] = sorted(work, key=lambda w: w["index"])
        self.meta["mesh_s"] = time.time() - self.t0
        return self.meta


def build_workload_2d(nb, pool=None, wait=True):
    """BASELINE configs[1]: Benchmark model 1 (2D axisymmetric), tool A0.4M6.0N, 100 depths in batches of 5, default mesh
    scale (the one that meets the reference's logs): nb of the 20 batches, evenly spread over the log."""
    m, depths, sim, batches, mud, bg = _model_and_batches_2d()
    mine = list(range(0, len(batches), max(1, len(batches) // nb)))[:nb]
    meta = dict(model=m, depths=depths, n_batches=len(batches), names=["A0.4M6.0N"])   # the slab of results is indexed by the depth's place in the whole log
    if pool is not None and len(mine) > 1:
        pend = Pending(meta, [pool.submit(_build_some_2d, [i]) for i in mine], time.time())
        return pend.result() if wait else pend
    meta["work"] = _build_some_2d(mine)
    return meta


def build_workload(rank, world, depths_per_gpu, scale, dim=3, mesh_3d="lattice", total_depths=None, all_batches=False, max_batches=None,
                   pool=None, wait=True):
    """Batches of this rank (block-cyclic share; all of them when all_batches) with meshes and right-hand sides.
    pool: a concurrent.futures executor of CPU-only processes that build the meshes side by side."""
    n_depths = int(total_depths) if total_depths else depths_per_gpu * world
    m, depths, sim, batches, mud, bg = _model_and_batches(n_depths)
    mine = list(range(len(batches))) if all_batches else list(range(rank, len(batches), world))
    if max_batches:
        mine = mine[:max_batches]
    t0 = time.time()
    meta = dict(model=m, depths=depths, n_batches=len(batches), names=list(TOOLS))
    if pool is not None and len(mine) > 1:
        order = mine
        if mesh_3d == "lattice":
            # a lattice mesh depends on the electrode pattern only (six distinct meshes in the 100-depth sweep, shared through the
            # on-disk cache of meshgen.cached_mesh by batches, ranks and reruns): the first batch of every pattern goes to the pool
            # first, so that all distinct meshes are in the making at once and the other jobs find them in the cache
            from remo3d_amd.model import lattice_mesh_key
            seen, first, rest = set(), [], []
            for i in mine:
                k = lattice_mesh_key(3, 50.0, batches[i], scale, 0)
                (rest if k in seen else first).append(i)
                seen.add(k)
            order = first + rest
            meta["distinct_meshes"] = len(first)
        pend = Pending(meta, [pool.submit(_build_some, (n_depths, scale, mesh_3d, [i])) for i in order], t0)
        return pend.result() if wait else pend
    meta["work"] = _build_some((n_depths, scale, mesh_3d, mine))
    meta["mesh_s"] = time.time() - t0
    return meta


def _cpu_leg_main(path, idx):
    """Child process of the CPU leg = ONE worker of the reference's farm (remo3d.py:592-595, worker.py:104-112): a single-threaded
    process that takes right-hand side `idx` of the sample and assembles and solves it on its own through oracle/fem_oracle.c -
    assembly, Jacobi-PCG to the same rtol, evaluation; result as JSON next to the input."""
    import pickle
    import numpy as np   # noqa: F401
    with open(path, "rb") as f:
        job = pickle.load(f)
    from oracle.fem_oracle import lib, solve_batch
    lib()                                            # compile / load outside the timed span
    bi, k = job["rhs"][idx]
    w = job["batches"][bi]
    z, I = w["sources"][k]
    ez = list(w["evals"][k])
    t0 = time.time()
    out, rc, st = solve_batch(w["mesh"], w["sigma"], [0, len(z)], list(z), list(I), [0, len(ez)], ez, condense=True, rtol=job["rtol"], maxit=job["maxit"])
    dt = time.time() - t0
    with open("%s.%d.json" % (path, idx), "w") as f:
        json.dump(dict(seconds=dt, rc=int(rc), iterations=st["iterations"], n=st.get("n"), out=[float(v) for v in out], batch=bi, rhs=k), f)


def under_profiler():
    """True when a rocprofv3 tool library is preloaded into this process: its GPU is initialised before main() runs, so no child
    process may be started from it (bench.py and the tools build their meshes in-process then; the on-disk mesh cache, filled by an
    un-profiled run of the same command, makes that a file read)."""
    return any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY", "HSA_TOOLS_LIB"))


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline_run(work, rtol, maxit, max_procs=10, timeout_s=900.0):
    """SURVEY 8d's CPU baseline with the oracle in NGSolve's place (NGSolve is not installable here): N single-threaded worker
    processes on N host cores, each assembling and solving one right-hand side of the headline workload on its own - the
    reference's one-process-per-worker farm (remo3d.py:592-595) - over the right-hand sides of the first whole batches (N = the
    cores this process may use, at most `max_procs` = two batches of five).  Runs AFTER the GPU legs (nothing is timed beside it).
    Returns (record, {(batch, rhs): potentials}) for the spot check of the GPU results against the oracle."""
    import pickle
    import tempfile
    cores = host_cores()
    n_proc = max(1, min(cores, max_procs if cores >= max_procs + 2 else 5))     # whole batches of five; two cores stay free for the parent
    rhs = [(bi, k) for bi, w in enumerate(work[:2]) for k in range(len(w["sources"]))][:n_proc]
    n_proc = len(rhs)
    fd, path = tempfile.mkstemp(prefix="remo_cpu_leg_", suffix=".pkl")
    with os.fdopen(fd, "wb") as f:
        pickle.dump(dict(batches=[dict(mesh=w["mesh"], sigma=w["sigma"], sources=w["sources"], evals=w["evals"]) for w in work[:2]], rhs=rhs,
                         rtol=rtol, maxit=maxit), f)
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-leg", path, str(i)], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
             for i in range(n_proc)]
    failed = None
    for pr in procs:
        try:
            pr.wait(timeout=max(1.0, timeout_s - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            failed = "a CPU worker did not finish within %.0f s" % timeout_s
            break
    wall = time.time() - t0
    for pr in procs:
        if pr.poll() is None:
            pr.kill()
    rec = dict(value=None, unit="points/s", cores=n_proc, kind="port", host_cores_available=cores)
    outs, secs, its, points, n_unknowns = {}, [], [], 0, None
    try:
        if failed is None:
            for i, (bi, k) in enumerate(rhs):
                try:
                    with open("%s.%d.json" % (path, i)) as f:
                        r = json.load(f)
                except OSError:
                    failed = "CPU worker %d failed: %s" % (i, procs[i].stderr.read().decode()[-300:])
                    break
                import numpy as np
                outs[(bi, k)] = np.asarray(r["out"]); secs.append(r["seconds"]); its.append(r["iterations"]); n_unknowns = r.get("n")
                points += len(work[bi]["readers"][k])
    finally:
        for q in [path] + ["%s.%d.json" % (path, i) for i in range(n_proc)]:
            try:
                os.remove(q)
            except OSError:
                pass
    if failed is not None:
        rec["sample"] = failed
        return rec, {}
    rec.update(value=points / wall, seconds_wall=wall, per_core_points_per_s=points / wall / n_proc, worker_seconds=secs, pcg_iterations=its,
               sample=f"{n_proc} right-hand sides ({points} points) of the headline workload = the first {len(set(b for b, _ in rhs))} batch(es), "
                      f"{n_unknowns} unknowns each, on {n_proc} single-threaded worker processes ({cores} host cores available to the bench), one right-hand side per "
                      f"worker as the reference farms them out (remo3d.py:592-595, worker.py:104-112: assembly + solve per right-hand side), started after the GPU legs: "
                      f"oracle/fem_oracle.c assembly + Jacobi-PCG (rtol as the GPU run, {min(its)}-{max(its)} steps) + evaluation, {min(secs):.1f}-{max(secs):.1f} s per worker, "
                      f"{wall:.1f} s wall.  NGSolve is not installable here, so this is the build's scalar C restatement, not the reference binary")
    return rec, outs


def box_stream(device):
    """What THIS box streams from HBM (remo_debug_stream): a 1 GiB read through a plain summing kernel and a 1 GiB device-to-device
    copy, HIP events, best of six.  The SpMM's time differs by up to 20 % between boxes of the pool; this says how much is the box."""
    import ctypes as C
    from remo3d_amd import _lib, solver
    try:
        with solver.Context(device) as ctx:
            r, c = C.c_double(0), C.c_double(0)
            rc = _lib.load().remo_debug_stream(ctx._h, 1 << 30, C.byref(r), C.byref(c))
            if rc != 0:
                return dict(error=ctx.last_error())
            g, m, mc = C.c_double(0), C.c_double(0), C.c_double(0)
            _lib.load().remo_debug_clock(ctx._h, C.byref(g))
            _lib.load().remo_debug_stream(ctx._h, 128 << 20, C.byref(m), C.byref(mc))   # fits the Infinity Cache: re-read back to back
            dev = (C.c_int64 * 8)()
            _lib.load().remo_debug_device(ctx._h, dev)
            lg, fg = C.c_double(0), C.c_double(0)
            _lib.load().remo_debug_cache_gather(ctx._h, 2 << 20, C.byref(lg))
            _lib.load().remo_debug_cache_gather(ctx._h, 256 << 20, C.byref(fg))
            xcc = (C.c_int32 * 1024)()
            _lib.load().remo_debug_xcc(ctx._h, xcc, 1024)
            xl = [int(v) for v in xcc]
            round_robin = all(xl[b] == (xl[0] + b) % 8 for b in range(1024))
            return dict(stream_read_GBs=r.value, stream_copy_GBs=c.value, infinity_cache_reread_GBs=m.value, l2_scattered_16B_reads_GBs=lg.value, scattered_16B_reads_over_256MiB_GBs=fg.value, dependent_fma_G_per_s_per_wave=g.value,
                        xcd_of_workgroups_0_to_15=xl[:16], xcd_is_workgroup_mod_8=bool(round_robin),
                        xcd_histogram=[xl.count(v) for v in range(8)],
                        device=dict(compute_units=dev[0], clock_MHz=dev[1] / 1e3, memory_clock_MHz=dev[2] / 1e3, bus_bits=dev[3], l2_bytes=dev[4], memory_MiB=dev[5], lds_bytes_per_cu=dev[6], revision=dev[7]),
                        note="1 GiB, 16-byte loads in a summing kernel / hipMemcpy device to device (read + write), best of 6; 128 MiB re-read four times back to back (Infinity Cache); chain of dependent fp32 "
                             "multiply-adds per wave, 1024 waves at once (follows the shader clock under load)")
    except Exception as ex:   # the probe is context, never a reason to lose the line
        return dict(error="%s: %s" % (type(ex).__name__, ex))


def pmc_traffic(workload, op="csr"):
    """HBM bytes per operator application from the committed rocprofv3 --pmc passes of this workload's mesh size and operator
    (profiles/, collected with tools/collect_traffic.sh + tools/pmc_traffic.py on the first batches of the SAME sweep - the meshes
    come from the on-disk cache, so the profiled process needs no child processes); None when size or operator differ."""
    import re
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            p = json.load(f)
    except OSError:
        return None
    size = re.search(r"mesh size (\w+)", workload or "")
    if size and p.get("mesh_size") == size.group(1) and p.get("operator", "csr") == op:
        return dict(bytes=p["spmm"]["traffic_bytes_per_launch"], n_free=p.get("n_free"), mesh_T=p.get("mesh_T"),
                    algorithmic=p.get("algorithmic_bytes_per_launch_same_launches"), workload=p.get("workload"))
    return None


def self_launch(args, argv):
    """--gpus N without a torchrun environment: spawn the N ranks as fresh child processes (this parent has not imported
    torch nor touched the GPU), relay rank 0's JSON line, exit non-zero if any rank failed."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if r.returncode != 0 or line is None:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank launch failed (exit status {r.returncode})\n")
        raise SystemExit(r.returncode or 1)
    rec = json.loads(line)
    if rec.get("n_gpus") != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} ranks, the line says {rec.get('n_gpus')}\n")
        raise SystemExit(1)
    print(line)


class Runner:
    """The timed part: the batches of one rank on one set of contexts, one pass = one_step().
    h2d_inclusive (the headline span): every batch goes through remo_solve_batch - host arrays to the device, run,
    potentials back - inside the pass; otherwise the batches are made resident first (remo_batch_create) and a pass runs them."""

    def __init__(self, work, n_depths, local, opts, streams=1, schedule="static", all_resident=False, resident=True, per_batch_opts=None):
        import numpy as np
        from remo3d_amd import solver, sweep, tasks
        self.np, self.sweep, self.tasks = np, sweep, tasks
        self.work, self.n_depths, self.opts, self.schedule = work, n_depths, opts, schedule
        self.per_batch_opts = per_batch_opts          # conforming meshes: the Chebyshev degree Model picks per mesh
        self.ctxs = [solver.Context(local) for _ in range(max(1, streams))]
        # dynamic schedule (all_resident: the rank holds the host data of every batch, the queue index is the batch index): nothing
        # is uploaded ahead - a drawn batch is brought in by an uploader context on its own stream while the one before it runs
        self.all_resident = all_resident
        self.uploader = solver.Context(local) if all_resident else None
        self.resident = []
        if resident and not all_resident:
            self.resident = [self.ctxs[i % len(self.ctxs)].batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for i, w in enumerate(work)]
        self.first_outs = {}                          # potentials of batches 0 and 1 of the last pass (spot check against the oracle)
        self.pool = None
        if len(self.ctxs) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=len(self.ctxs))

    def close(self):
        for b in self.resident:
            b.close()
        for c in self.ctxs + ([self.uploader] if self.uploader else []):
            c.close()
        if self.pool is not None:
            self.pool.shutdown()

    def _opts(self, i):
        return self.per_batch_opts[i] if self.per_batch_opts else self.opts

    def _collect(self, i, rc, st, outs, slab, agg):
        np = self.np
        w = self.work[i]
        if rc < 0:
            for rd in w["readers"]:
                for (di, ti, K, o, m) in rd:
                    slab[di, ti] = np.nan
            return
        agg["not_converged"] += int(rc == 1)
        if i < 2:
            self.first_outs[i] = [u.copy() for u in outs]
        for u, rd in zip(outs, w["readers"]):
            for (di, ti, K, o, m) in rd:
                slab[di, ti] = self.tasks.apparent_resistivity(u[o:o + m], m, K, w["mesh"].dim)
        agg["spmv_ms"] += st["spmv_ms"]; agg["spmv_launches"] += st["spmv_launches"]
        agg["spmv_ms_raw"] += st["spmv_ms_raw"]; agg["ev_over"] = st["event_overhead_ms"]
        agg["spmv_bytes_total"] += st["spmv_bytes"] * st["spmv_launches"]
        krhs = max(1, len(w["sources"]))
        agg["spmv_flops_total"] = agg.get("spmv_flops_total", 0.0) + st["spmv_launches"] * float(
            PATCH_FLOPS_PER_ELEMENT_COLUMN * w["mesh"].n_elems * krhs if st["op_used"] == 3 else 2.0 * st["nnz"] * krhs)
        agg["pcg_steps"] += st["pcg_steps"]; agg["max_it"] = max(agg["max_it"], st["max_iterations"])
        for k in ("ms_symbolic", "ms_assemble", "ms_solve", "ms_h2d", "ms_eval"):
            agg[k] += st[k]
        agg["n"] = st["n_free"]; agg["nnz"] = st["nnz"]; agg["batches"] += 1; agg["op_used"] = st["op_used"]; agg["coarse_used"] = st["coarse_used"]
        if i == 0:
            agg["n0"] = st["n_free"]; agg["T0"] = w["mesh"].n_elems

    def one_step(self, h2d_inclusive=False):
        np = self.np
        slab = np.zeros((self.n_depths, len(TOOLS)))
        agg = dict(spmv_ms=0.0, spmv_ms_raw=0.0, spmv_launches=0, spmv_bytes_total=0.0, pcg_steps=0, not_converged=0, ms_symbolic=0.0,
                   ms_assemble=0.0, ms_solve=0.0, ms_h2d=0.0, ms_eval=0.0, n=0, nnz=0, max_it=0, batches=0, ev_over=0.0, op_used=0)
        t_busy = time.time()
        nctx = len(self.ctxs)
        if h2d_inclusive:     # the host-buffer entry: every batch is copied to the device inside the timed span (remo_solve_batch)
            # as Model.simulate_logs drives its contexts: whichever context is free takes the rank's next batch (REMO_BENCH_CTX_DRAW=static:
            # context j takes batches j, j + nctx, ... - the rule before, kept for the A/B)
            import itertools, threading
            draw, draw_lock = itertools.count(), threading.Lock()
            static_draw = os.environ.get("REMO_BENCH_CTX_DRAW") == "static"

            def my_batches(j):
                if static_draw:
                    yield from range(j, len(self.work), nctx)
                    return
                while True:
                    with draw_lock:
                        i = next(draw)
                    if i >= len(self.work):
                        return
                    yield i

            def drive_h(j):
                res = []
                for i in my_batches(j):
                    w = self.work[i]
                    outs, st, rc = self.ctxs[j].solve_batch(w["mesh"], w["sigma"], w["sources"], w["evals"], self._opts(i), raise_on_error=False)
                    res.append((i, rc, st, outs))
                return res
            chunks = self.pool.map(drive_h, range(nctx)) if self.pool is not None else [drive_h(0)]
            for i, rc, st, outs in sorted((p for chunk in chunks for p in chunk), key=lambda p: p[0]):
                self._collect(i, rc, st, outs, slab, agg)
        elif self.all_resident:      # pull scheduling: draw, upload (one batch ahead, on the uploader's stream), run, release
            from concurrent.futures import ThreadPoolExecutor
            queue = iter(self.sweep.BatchQueue(len(self.work), self.schedule))

            def bring(i):
                w = self.work[i]
                return self.uploader.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
            with ThreadPoolExecutor(max_workers=1) as up:
                i = next(queue, None)
                fut = up.submit(bring, i) if i is not None else None
                while fut is not None:
                    cur_i, cur = i, fut.result()
                    i = next(queue, None)            # a drawn batch is OWNED: one ahead, no more (the other ranks may be idle for it)
                    fut = up.submit(bring, i) if i is not None else None
                    rc = cur.run(self._opts(cur_i), raise_on_error=False, ctx=self.ctxs[0])
                    self._collect(cur_i, rc, cur.stats, cur.fetch() if rc >= 0 else None, slab, agg)
                    cur.close()
        else:                        # resident batches: one host thread per context, each walks its own batches in order (ctypes releases the GIL)
            def drive(j):
                res = []
                for i in range(j, len(self.resident), nctx):
                    rc = self.resident[i].run(self._opts(i), raise_on_error=False)
                    res.append((i, rc, self.resident[i].stats, self.resident[i].fetch() if rc >= 0 else None))
                return res
            chunks = self.pool.map(drive, range(nctx)) if self.pool is not None else [drive(0)]
            for i, rc, st, outs in sorted((p for chunk in chunks for p in chunk), key=lambda p: p[0]):
                self._collect(i, rc, st, outs, slab, agg)
        agg["busy_s"] = time.time() - t_busy
        slab = self.sweep.combine(slab)   # the ONE collective of the path: all-reduce of the log slab
        return slab, agg


def timed(runner, steps, warmup, sync, **kw):
    for _ in range(warmup):
        runner.one_step(**kw)
    sync()
    t0 = time.time()
    busy = 0.0
    for _ in range(steps):
        slab, agg = runner.one_step(**kw)
        busy += agg["busy_s"]
    sync()
    return time.time() - t0, slab, agg, busy


# fp64 work of the patch operator per (element, right-hand side): 427 v_fmac_f64 + 90 v_add_f64 + 83 v_mul_f64 in the ISA of
# k_patch_apply<double, 5, 256> (tools/kernel_resources.py leaves it in /tmp; counters: profiles/r03_pmc_sq_counters_patch_L_final.txt)
PATCH_FLOPS_PER_ELEMENT_COLUMN = 2 * 427 + 90 + 83
# MI355X fp64 vector rate = its fp64 matrix rate: 256 CUs x 4 SIMDs x 16 FMA lanes x 2 x 2.4 GHz (half the guide's fp32 figure)
FP64_PEAK_TFLOPS = 78.6


def roofline_of(agg, precision, stride, workload_name=None, contexts=1):
    """The operator application of the CG (what `time_kernels` brackets: every launch of it) against the HBM roofline, priced by
    ITS OWN algorithmic bytes (remo_stats_t.spmv_bytes): CSR product 12 nnz + 4 n + 16 k n; patch operator 16 k n + 88 T (x read and
    y written once, 40 bytes of local indices + 48 of metric terms per element - no stored entries)."""
    ach = (agg["spmv_bytes_total"] / 1e9) / (agg["spmv_ms"] / 1e3) if agg["spmv_ms"] > 0 else None
    op = {0: "csr", 3: "patch"}.get(int(agg.get("op_used", 0)), "csr")
    tr = pmc_traffic(workload_name, op) if (workload_name and precision == "fp64") else None
    r = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=(ach / HBM_PEAK_GBS) if ach else None,
             traffic=tr["bytes"] if tr else None)
    if tr:     # the counter passes ran on the first batches of the SAME sweep (meshes from the on-disk cache): compare like with like
        alg = tr.get("algorithmic") or ((16.0 * 5 * tr["n_free"] + 88.0 * tr["mesh_T"]) if op == "patch" and tr.get("n_free") and tr.get("mesh_T") else None)
        r["traffic_measured_on"] = dict(workload=tr.get("workload"), algorithmic_bytes_there=alg,
                                        traffic_over_algorithmic=(tr["bytes"] / alg) if alg else None, file=PMC_FILE,
                                        note="committed rocprofv3 --pmc passes (tools/collect_traffic.sh), not measured in this run")
    prec = "fp64" if precision == "fp64" else "fp32 values and vectors"
    kernel = {"csr": "k_spmm_pair (CSR SpMM, %s, k=5 interleaved RHS)",
              "patch": "k_patch_apply (matrix-free patch operator, %s, k=5 interleaved RHS; inside the PCG the rows shared by several patches are summed by the update launch and <p, A p> is added by the patches themselves: one launch per application)"}[op] % prec
    fp = precision == "fp64"
    formula = {"csr": "12*nnz + 4*n + 16*k*n (SURVEY.md 8d)" if fp else "8*nnz + 4*n + 8*k*n (SURVEY.md 8d, fp32 storage)",
               "patch": "16*k*n + 88*T (x and y once, 40 B local indices + 48 B metric terms per element)" if fp else "8*k*n + 88*T"}[op]
    r.update(kernel=kernel, operator=op, contexts_on_the_gpu=contexts,
             timed="every %d-th application of every solve, HIP events on the solver's stream" % stride,
             launches=int(agg["spmv_launches"]),
             avg_launch_us=(1e3 * agg["spmv_ms"] / agg["spmv_launches"]) if agg["spmv_launches"] else None,
             avg_bracket_us_raw=(1e3 * agg["spmv_ms_raw"] / agg["spmv_launches"]) if agg["spmv_launches"] else None,
             empty_event_pair_us=1e3 * agg["ev_over"],
             bytes_per_launch=formula,
             algorithmic_bytes_per_launch=(agg["spmv_bytes_total"] / agg["spmv_launches"]) if agg["spmv_launches"] else None,
             traffic_unit="bytes per application: reads sized by the TCC_EA0_RDREQ 32/64/128-B request counters + WRITE_SIZE over the kernels of the bracket")
    # the compute side of the same launches: a matrix-free operator trades bytes for arithmetic, and above ~10 flop per byte
    # (78.6 TFLOP/s over 8 TB/s) the fp64 rate is the roof that binds, not HBM
    fl = agg.get("spmv_flops_total", 0.0)
    if fl > 0 and agg["spmv_ms"] > 0 and precision == "fp64":
        tf = fl / (agg["spmv_ms"] / 1e3) / 1e12
        r["compute"] = dict(flops_per_launch=fl / agg["spmv_launches"], achieved_tflops=tf, peak_tflops=FP64_PEAK_TFLOPS, frac=tf / FP64_PEAK_TFLOPS,
                            flop_per_algorithmic_byte=fl / agg["spmv_bytes_total"], machine_balance_flop_per_byte=FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS,
                            note="fp64 vector (= matrix) rate of MI355X; flops counted from the kernel's ISA (patch operator) or 2 nnz k (CSR product)")
    return r


def model_end_to_end(n_depths, cpu_workers):
    """BASELINE configs[2] through the product's front door: Model.compute_synthetic_logs on Benchmark model 3 (dip 30), both tools,
    `n_depths` depths - interface-conforming meshes at the default scale built by `cpu_workers` mesh processes AHEAD of the solver,
    Model's default number of contexts on the GPU, uploads, solves, apparent resistivities.  Everything a user waits for is inside the span."""
    import numpy as np
    from remo3d_amd.model import Model
    ex = os.path.join(ROOT, "tests", "golden", "examples", "Benchmark models", "Benchmark model 3")
    depths = np.linspace(5.0, 20.0, n_depths, endpoint=False)
    t0 = time.time()
    m = Model.compute_synthetic_logs(TOOLS, depths, os.path.join(ex, "Formation_BM3_30.txt"), os.path.join(ex, "Borehole_BM3.txt"), dip=30,
                                     cpu_workers=cpu_workers, verbose=False)
    dt = time.time() - t0
    t = m.timing
    pts = n_depths * len(TOOLS)
    nan = int(sum(np.isnan(v[:, 1]).sum() for v in m.logs.values()))
    return dict(value=pts / dt, unit="points/s", points=pts, seconds=dt, includes="mesh generation (interface-conforming revolved meshes, scale 1.0, %d mesh processes ahead of the solver), "
                "host -> device, numbering, assembly, PCG, evaluation, device -> host, apparent resistivity; Model's default number of contexts on the GPU" % min(int(cpu_workers), 8),
                batches=t.get("batches"), pcg_steps_per_batch=(t.get("pcg_steps", 0) / max(1, t.get("batches", 1))), busy_s=t.get("busy_s"),
                sum_of_mesh_waits_s=t.get("mesh_s"), sum_of_solve_calls_s=t.get("solve_s"), failed_batches=t.get("failed_batches"), nan_points=nan)


def log(msg):
    """Progress on stderr (stdout carries the ONE JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - T_START, msg))
        sys.stderr.flush()


T_START = time.time()


def _tuned(d, forced="auto"):
    """Model's per-mesh choice for conforming meshes as keywords of main()'s make_opts (`coarse` travels as coarse_solver)."""
    d = dict(d)
    c = d.pop("coarse", "auto")
    d["coarse_solver"] = forced if forced != "auto" else c
    return d


def size_scale(size):
    """'S' / 'M' / 'L' / 'XL' or a number = the multiplier on the reference's size field itself."""
    return SIZES[size] if size in SIZES else float(size)


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-leg":
        return _cpu_leg_main(sys.argv[2], int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", default="L", choices=list(SIZES),
                    help="multiplier class of the reference size field: L = 1.2 x (the reference's resolution, ~2 M dofs per batch: the headline), "
                         "S = 4 x coarser (the round-1/2 headline), M, XL (BASELINE config 5's ~5 M dofs)")
    ap.add_argument("--depths", type=int, default=100, help="measurement depths per GPU (weak scaling)")
    ap.add_argument("--total-depths", type=int, default=0,
                    help="strong scaling: this many depths in all, shared by the ranks (BASELINE configs[3]: 1000)")
    ap.add_argument("--batches", type=int, default=0, help="profiling runs: only the first N batches of the rank's share")
    ap.add_argument("--schedule", default="static", choices=["static", "dynamic"],
                    help="static = block-cyclic shares; dynamic = ranks draw batches from a shared counter while they are free "
                         "(the reference's pull scheduling); a drawn batch is uploaded by a second context while the one before it runs")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--maxsteps", type=int, default=1000)
    ap.add_argument("--mesh", default="lattice", choices=["lattice", "conforming"],
                    help="lattice = the seeded synthetic half-ball meshes of SURVEY 8d (headline workload); conforming = the interface-"
                         "conforming revolved meshes Model uses by default for dipping models")
    ap.add_argument("--precision", default="fp64", choices=["fp64", "mixed"],
                    help="fp64 (the headline configuration) or mixed = fp32 PCG inside fp64 refinement (BASELINE config 5)")
    ap.add_argument("--op", default="auto", choices=["auto", "csr", "patch"],
                    help="how the CG applies A: csr = SpMM on the assembled matrix; patch = matrix-free through the factorised reference tensors, "
                         "patch by patch with LDS-staged vectors; auto (the library's default) = patch in 3D")
    ap.add_argument("--streams", type=int, default=5,
                    help="contexts (HIP streams + arenas) per GPU, each driven by its own host thread over its share of the batches; "
                         "3 = the headline configuration and Model's default, model.DEFAULT_CONTEXTS (the launch-latency-bound part of one batch's PCG step is filled by the others' kernels)")
    ap.add_argument("--resident", action="store_true",
                    help="headline leg with the batches resident on the device before the timed region (remo_batch_create) instead of SURVEY 8d's span "
                         "(host arrays -> device inside it); per-kernel profiling runs use this with --streams 1")
    ap.add_argument("--overlap", default="all", choices=["prepare", "all"],
                    help="with --streams > 1: 'prepare' = only one batch is in its PCG at a time, the other contexts number / assemble "
                         "theirs beside it; 'all' = no restriction (default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--sizes", default="L/csr:4,L/mixed:4,S:8,M:8,XL:2,XL/mixed:2,conforming-M:8,conforming-1.0:4,2D-BM1:8",
                    help="further workloads measured in the same run at N = 1 (SIZE:batches; SIZE = S / M / L / XL or a number = the multiplier itself; 'conforming-' prefix = "
                         "the interface-conforming meshes Model builds, with Model's Chebyshev degree per mesh (conforming-1.0 = Model's default scale), "
                         "'/mixed' = fp32 PCG in fp64 refinement, '/csr' = that operator instead of the choice by size; '2D-BM1' = BASELINE configs[1], "
                         "Benchmark model 1 in 2D, '/chebyshev' = polynomial instead of the multigrid cycle on the vertex block, '/2ctx' = two contexts (streams, host threads) share the batches), reported in the `sizes` array; '' = none")
    ap.add_argument("--no-extras", action="store_true", help="only the headline leg (no kernel-timing leg, sizes, end-to-end or CPU legs)")
    ap.add_argument("--mesh-workers", type=int, default=10, help="CPU processes that build the synthetic meshes side by side (before any GPU work)")
    ap.add_argument("--coarse", default="", metavar="DEGREE,RATIO", help="experiments only: Chebyshev degree and interval ratio of the P1 block (default: by vertex count)")
    ap.add_argument("--vertex-solver", default="auto", choices=["auto", "chebyshev", "amg", "amg_or_chebyshev"],
                    help="experiments only: solver of the P1 block inside the two-level preconditioner (auto = the library's default: polynomial in 3D)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B experiments only (library built with -DREMO_PROBES): remo_debug_tune(KEY, VALUE) before the run (include/remo3d_hip_debug.h lists the keys)")
    ap.add_argument("--no-events", action="store_true", help="do not bracket SpMV launches with HIP events")
    ap.add_argument("--event-stride", type=int, default=8,
                    help="bracket every k-th SpMV launch of a solve with HIP events (a bracket costs the stream ~1.5 us: bracketing "
                         "every launch takes 6 %% off the throughput it is there to explain)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args, sys.argv[1:])        # nothing of torch / HIP has been loaded in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("REMO_DEVICE", os.environ.get("LOCAL_RANK", "0")))   # REMO_DEVICE: rehearsals on a 1-GPU box
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a different number of ranks")

    import numpy as np
    strong = args.total_depths > 0
    dynamic = args.schedule == "dynamic" and world > 1
    profiled = under_profiler()
    extras = ((world == 1) and not args.no_extras and not args.resident and args.precision == "fp64" and args.mesh == "lattice" and not args.tune
              and args.op == "auto" and not args.coarse and args.vertex_solver == "auto" and not args.batches and not profiled)
    extra_specs = []
    if extras and args.sizes:
        for spec in args.sizes.split(","):
            name, nb = spec.split(":")
            conf = name.startswith("conforming-")
            size = name.split("/")[0].split("-", 1)[-1] if conf else name.split("/")[0]
            if name.startswith("2D"):
                extra_specs.append((name, "2D", "2D", int(nb)))
            elif not (name == args.size):
                extra_specs.append((name, size, "conforming" if conf else "lattice", int(nb)))

    # ---- synthetic meshes: CPU-only worker processes, before this process loads the HIP library ----
    mesh_pool = None
    if profiled:
        log("running under a profiler: meshes are built in this process (no child processes behind a preloaded tool library; "
            "an un-profiled run of the same command before it leaves them in the on-disk cache)")
    # every rank builds the meshes of its own batches in its own CPU-only worker processes (N ranks share the node's cores; the
    # distinct lattice meshes are shared between ranks through the on-disk cache: whoever comes first builds, the others load)
    cores = host_cores()
    n_workers = args.mesh_workers if world == 1 else max(2, min(args.mesh_workers, cores // world))
    if n_workers > 1 and not profiled:
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor
        mesh_pool = ProcessPoolExecutor(max_workers=n_workers, mp_context=multiprocessing.get_context("spawn"))
    t_mesh0 = time.time()
    if mesh_pool is not None:
        os.environ.setdefault("OMP_NUM_THREADS", "1")      # inherited by the spawned mesh workers: one thread each
    # every workload of the run is SUBMITTED before any is waited for; XL first (its meshes take longest)
    pending = {}
    for name, size, kind, nb in sorted(extra_specs, key=lambda e: -{"XL": 3, "L": 2, "M": 1}.get(e[1], 0)):     # the 20-depth sweep of the same model; variants of one size share its meshes
        if (size, kind, nb) not in pending and not (size == args.size and kind == args.mesh):
            pending[(size, kind, nb)] = (build_workload_2d(nb, pool=mesh_pool, wait=False) if kind == "2D" else
                                         build_workload(0, 1, 20, size_scale(size), mesh_3d=kind, max_batches=nb, pool=mesh_pool, wait=False))
    wl = build_workload(rank, world, args.depths, SIZES[args.size], mesh_3d=args.mesh, total_depths=args.total_depths or None,
                        all_batches=dynamic, pool=mesh_pool, wait=False, max_batches=args.batches or None)

    def progress(what):
        return lambda done, total: log("meshes: %s %d / %d" % (what, done, total)) if (done % 5 == 0 or done == total) else None
    all_futs = [f for pend in list(pending.values()) + [wl] if isinstance(pend, Pending) for f in pend.futs]
    import threading
    stop_beat = threading.Event()

    def beat():       # the mesh phase of the reference-resolution workload can take a minute: say so while it lasts
        while not stop_beat.wait(30.0):
            log("meshing: %d of %d mesh jobs done" % (sum(f.done() for f in all_futs), len(all_futs)))
    if all_futs:
        threading.Thread(target=beat, daemon=True).start()
    if isinstance(wl, Pending):
        wl = wl.result(progress("headline workload"))
    built = {}
    for key, pend in pending.items():
        built[key] = pend.result(progress("%s (%s)" % (key[0], key[1]))) if isinstance(pend, Pending) else pend
    extra_wl = []
    for name, size, kind, nb in extra_specs:
        if size == args.size and kind == args.mesh:     # variants of the headline size run on the headline's own first batches
            w2 = dict(wl)
            w2["work"] = wl["work"][:nb]
            extra_wl.append((name, w2))
        else:
            extra_wl.append((name, built[(size, kind, nb)]))
    stop_beat.set()
    if mesh_pool is not None:
        mesh_pool.shutdown()
    mesh_wall = time.time() - t_mesh0
    log("meshes built: %d batches of the headline workload + %s in %.1f s" % (len(wl["work"]), [(n, len(w["work"])) for n, w in extra_wl], mesh_wall))

    from remo3d_amd import solver, sweep
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

    stride = max(1, args.event_stride)

    def make_opts(time_kernels=True, precision=args.precision, op=args.op, coarse="auto", streams=args.streams, **kw):
        coarse = kw.pop("coarse_solver", coarse)
        if coarse == "auto" and args.vertex_solver != "auto":
            coarse = args.vertex_solver
        if args.coarse:
            kw.setdefault("coarse_degree", int(args.coarse.split(",")[0])); kw.setdefault("coarse_ratio", int(args.coarse.split(",")[1]))
        return solver.make_opts(preconditioner="multigrid", condense=True, rtol=args.rtol, maxsteps=args.maxsteps,
                                time_kernels=stride if (time_kernels and not args.no_events) else 0, precision=precision,
                                serialize_solves=(streams > 1 and args.overlap == "prepare"), op=op, coarse=coarse, **kw)
    opts = make_opts()
    work = wl["work"]
    n_depths = len(wl["depths"])
    # the solver of the P1 block is the one `Model` would ask for on these meshes with this many contexts (model.vertex_solver_options):
    # interface-conforming meshes - the multigrid cycle with the tuned polynomial behind it; lattice meshes - the cycle when several contexts
    # share the GPU (the headline), the library's default polynomial on one context (the kernel-timing leg and the `sizes` legs)
    from remo3d_amd.model import vertex_solver_options

    def per_batch_opts(batches, n_ctx, conforming, **kw):
        if args.vertex_solver != "auto" or args.coarse:
            return None
        return [make_opts(streams=n_ctx, **kw, **_tuned(vertex_solver_options(w["mesh"].dim, w["mesh"].n_nodes, conforming, n_ctx))) for w in batches]
    per_batch = per_batch_opts(work, args.streams, args.mesh == "conforming")
    runner = Runner(work, n_depths, local, opts, streams=args.streams, schedule=args.schedule, all_resident=dynamic, resident=args.resident,
                    per_batch_opts=per_batch)
    h2d = not args.resident and not dynamic

    def sync():
        if dist_on:
            import torch
            sweep.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    log("headline leg (%d context(s), %s): timing %d + %d steps" % (args.streams, "host -> device inside the span" if h2d else "batches resident / drawn", args.warmup, args.steps))
    dt_local, slab, agg, busy = timed(runner, args.steps, args.warmup, sync, h2d_inclusive=h2d)
    log("timed region done: %.3f s" % dt_local)
    dt = sweep.max_over_ranks(dt_local)
    busy_ranks = sweep.gather_floats([1e3 * busy / args.steps, float(agg["batches"])])
    first_outs = dict(runner.first_outs)
    runner.close()

    n_points = n_depths * len(TOOLS) if not args.batches else sum(len(rd) for w in work for rd in w["readers"])
    value = n_points * args.steps / dt
    per = f"{args.total_depths} depths in all" if strong else f"{args.depths} depths/GPU"
    workload_name = f"BM3 dip30, tools A0.4M6.0N+A2.0M0.5N, {per}, R=50, batch 5, mesh size {args.size}"
    if args.mesh != "lattice":
        workload_name += ", interface-conforming revolved meshes"
    if args.batches:
        workload_name += f", first {args.batches} batches only"
    if rank != 0:
        return
    span = ("per batch: host arrays -> device, numbering, assembly, PCG, evaluation, potentials -> host (remo_solve_batch; SURVEY 8d's span), then one all-reduce; mesh generation excluded"
            if h2d else "batches resident on the device before the timed region (remo_batch_create): numbering, assembly, PCG, evaluation, fetch, one all-reduce")
    out = dict(metric="measurement points/sec (3D benchmark model)", value=value, unit="points/s", n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="strong" if strong else "weak",
               vs_baseline=None, dtype="f64" if args.precision == "fp64" else "f32 PCG inside f64 residual refinement", data="synthetic",
               config=dict(workload=workload_name, timed_span=span, contexts_per_gpu=args.streams, schedule=args.schedule if world > 1 else "single rank",
                           batches_total=wl["n_batches"], batches_rank0=int(agg["batches"]), rhs_rank0=sum(len(w["sources"]) for w in work) if not dynamic else None,
                           points_total=n_points, mesh_T=int(work[0]["mesh"].n_elems), n_free=int(agg.get("n0", agg["n"])), nnz=int(agg["nnz"]), rtol=args.rtol,
                           maxsteps=args.maxsteps, precision=args.precision, operator={0: "csr", 3: "patch"}.get(int(agg["op_used"]), "csr"),
                           preconditioner="multigrid = solver of the P1 vertex block + Jacobi on edge/face dofs; block solver as Model chooses it (model.vertex_solver_options): one smoothed-aggregation multigrid cycle when several contexts share the GPU, "
                                          "else a Chebyshev polynomial (degree / interval by vertex count: 12 on lmax/320 at 83 k vertices)",
                           vertex_block_solver={0: "none", 1: "chebyshev", 2: "multigrid cycle"}[int(agg.get("coarse_used", 1))],
                           max_pcg_iterations=int(agg["max_it"]), pcg_steps_per_batch=agg["pcg_steps"] / max(1, agg["batches"]),
                           batches_not_converged=int(agg["not_converged"]), nan_points=int(np.isnan(slab).sum()),
                           distinct_meshes=wl.get("distinct_meshes"), mesh_generation_excluded_s=wl["mesh_s"], all_mesh_generation_wall_s=mesh_wall),
               per_rank=dict(busy_ms_per_step=[b[0] for b in busy_ranks], batches_last_step=[int(b[1]) for b in busy_ranks]))
    if args.tune:
        out["config"]["debug_tune"] = list(args.tune)

    def breakdown(a):
        return dict(numbering_and_pattern_device=a["ms_symbolic"], h2d=a["ms_h2d"], assemble=a["ms_assemble"], solve=a["ms_solve"],
                    eval=a["ms_eval"], pcg_steps=int(a["pcg_steps"]), note="sums over the batches of one step of the per-batch device times (HIP events)")

    if extras or (world == 1 and args.streams == 1):
        # ---- the kernel-timing leg: ONE context, batches resident - nothing else on the chip while a launch is bracketed ----
        if args.streams == 1 and args.resident:
            agg_k, dt_k, slab_k, k_steps = agg, dt, slab, args.steps            # the headline leg already is that leg
        else:
            r1 = Runner(work, n_depths, local, make_opts(streams=1), streams=1, per_batch_opts=per_batch_opts(work, 1, args.mesh == "conforming"))
            k_steps = max(1, min(args.steps, 2))
            dt_k, slab_k, agg_k, _ = timed(r1, k_steps, 1, sync)
            first_outs = first_outs or dict(r1.first_outs)
            r1.close()
            log("kernel-timing leg (one context, resident) done: %.3f s" % dt_k)
        out["roofline"] = roofline_of(agg_k, args.precision, stride, workload_name, contexts=1)
        out["roofline"]["measured_in"] = ("the headline leg" if agg_k is agg else
                                          "a single-context leg of the same batches, resident, %d step(s) right after the headline leg: kernels of several contexts share the chip, "
                                          "so the headline leg's own brackets (roofline.in_timed_region) time a launch beside the other context's kernels" % k_steps)
        if agg_k is not agg:
            rt = roofline_of(agg, args.precision, stride, None, contexts=args.streams)
            out["roofline"]["in_timed_region"] = dict(avg_launch_us=rt["avg_launch_us"], achieved=rt["achieved"], frac=rt["frac"], launches=rt["launches"],
                                                      contexts_on_the_gpu=args.streams)
            out["value_one_context_resident"] = dict(value=n_points * k_steps / dt_k, unit="points/s", steps=k_steps,
                                                     max_abs_log_diff_vs_headline=float(np.nanmax(np.abs(slab_k - slab))),
                                                     note="round 3's headline configuration: one context, inputs resident before the timed region")
        out["breakdown_ms_per_step"] = breakdown(agg_k)
        # assembly against ITS roofline (8 nnz values written once + 4 nnz column reads + T (4 (dim + 1) + 4) + 8 dim nv bytes)
        if agg_k["batches"] and agg_k["ms_assemble"] > 0:
            m0 = work[0]["mesh"]
            asm_bytes = 12.0 * agg_k["nnz"] + m0.n_elems * (4 * 4 + 4) + 8.0 * 3 * m0.n_nodes
            us = 1e3 * agg_k["ms_assemble"] / agg_k["batches"]
            out["assembly"] = dict(kernels="k_metric_terms + k_assemble (HIP events around both)", algorithmic_bytes_per_batch=asm_bytes, us_per_batch=us,
                                   achieved_GBs=asm_bytes / us / 1e3, frac_of_hbm_peak=asm_bytes / us / 1e3 / HBM_PEAK_GBS)
    else:
        out["roofline"] = roofline_of(agg, args.precision, stride, workload_name, contexts=args.streams)
        out["roofline"]["measured_in"] = "the headline leg itself (%d contexts share the chip: a bracketed launch runs beside the other context's kernels)" % args.streams
        out["breakdown_ms_per_step"] = breakdown(agg)

    if extras:
        out["box"] = box_stream(local)
    sizes = []
    for name, w2 in extra_wl:
        prec2 = "mixed" if "/mixed" in name else args.precision
        op2 = "csr" if "/csr" in name else ("patch" if "/patch" in name else args.op)
        coarse2 = "chebyshev" if "/chebyshev" in name else ("amg_or_chebyshev" if "/amg" in name else "auto")
        nctx2 = 2 if "/2ctx" in name else 1
        pb2 = None
        if not name.startswith("2D"):
            pb2 = [make_opts(precision=prec2, op=op2, streams=nctx2, **_tuned(vertex_solver_options(3, w["mesh"].n_nodes, name.startswith("conforming-"), nctx2), coarse2)) for w in w2["work"]]
        r2 = Runner(w2["work"], len(w2["depths"]), local, make_opts(precision=prec2, op=op2, coarse=coarse2, streams=nctx2), streams=nctx2, per_batch_opts=pb2)
        st2 = 2
        dt2, slab2, agg2, _ = timed(r2, st2, 1, sync)
        pts = sum(len(rd) for w in w2["work"] for rd in w["readers"])
        rf = roofline_of(agg2, prec2, stride)
        sizes.append(dict(workload=name, precision=prec2, operator=rf["operator"], vertex_block_solver={0: "none", 1: "chebyshev", 2: "multigrid cycle"}[agg2.get("coarse_used", 1)], contexts=nctx2, resident=True,
                          batches=len(w2["work"]), points=pts, value=pts * st2 / dt2, unit="points/s", mesh_T=int(w2["work"][0]["mesh"].n_elems),
                          n_free=int(agg2.get("n0", agg2["n"])), nnz=int(agg2["nnz"]), pcg_steps_per_batch=agg2["pcg_steps"] / max(1, agg2["batches"]),
                          max_pcg_iterations=int(agg2["max_it"]), operator_frac_of_hbm_peak=rf["frac"], operator_bytes=rf["bytes_per_launch"], apply_avg_launch_us=rf["avg_launch_us"],
                          solve_ms_per_batch=agg2["ms_solve"] / max(1, agg2["batches"]), nan_points=int(np.isnan(slab2).sum())))
        r2.close()
        log("size leg %s done: %.3f s for %d steps" % (name, dt2, st2))
    if extras:
        out["sizes"] = sizes
        try:
            out["model_end_to_end"] = model_end_to_end(args.depths, cpu_workers=min(8, max(2, cores - 2)))
            log("Model end-to-end leg done: %.1f s" % out["model_end_to_end"]["seconds"])
        except Exception as ex:      # context, never a reason to lose the line
            out["model_end_to_end"] = dict(error="%s: %s" % (type(ex).__name__, ex))
    out["cpu_baseline"] = None
    if extras and not args.no_cpu:
        cb, ref_outs = cpu_baseline_run(work, args.rtol, 20000)
        log("CPU baseline leg done")
        out["cpu_baseline"] = cb
        worst = [float(np.max(np.abs(first_outs[bi][k] - ref) / np.abs(ref))) for (bi, k), ref in ref_outs.items() if bi in first_outs]
        if worst:
            out["config"]["gpu_vs_oracle_max_rel_diff_of_the_cpu_sample"] = max(worst)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
