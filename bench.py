#!/usr/bin/env python3
"""bench.py — measurement points per second of the ReMo3D hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): the reference's
3D benchmark model — Examples/Benchmark models/Benchmark model 3, dip 30 deg — logged with a
normal and a lateral tool (A0.4M6.0N, A2.0M0.5N) at 100 depths per GPU in [5, 20) m, R = 50 m,
batch_size 5 (=> 40 batches / 200 right-hand sides / 200 measurement points per GPU, computed
with the build's task builder, pinned against the reference's in tests/golden/tasks_bm3.json).
Batch meshes are seeded synthetic half-ball meshes with the reference's size field (no Gmsh in
the image); their sizes (T, n, nnz) are printed in the JSON line.

One "step" = one pass of the hot path over every batch of the sweep: dof numbering, CSR pattern,
assembly, multi-RHS two-level PCG, axis evaluation, apparent resistivity — then ONE all-reduce of
the log slab across ranks (RCCL).  Mesh arrays, sigma and points are resident on the device before
the timed region (remo_batch_create); mesh generation is excluded, as SURVEY.md section 8d defines
the point.  The same line also carries: the rate with the per-batch host->device copy inside the
timed span (remo_solve_batch, `value_h2d_inclusive`), and the same workload at the larger mesh sizes and on
the interface-conforming meshes `Model` uses (`sizes`).

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
PMC_FILE = os.path.join("profiles", "r03_pmc_traffic_default_bench.json")


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
        pend = Pending(meta, [pool.submit(_build_some, (n_depths, scale, mesh_3d, [i])) for i in mine], t0)
        return pend.result() if wait else pend
    meta["work"] = _build_some((n_depths, scale, mesh_3d, mine))
    meta["mesh_s"] = time.time() - t0
    return meta


def _cpu_leg_main(path):
    """Child process of the CPU leg: ONE right-hand side of the workload (batch 0, RHS 0) through oracle/fem_oracle.c - assembly,
    Jacobi-PCG to the same rtol, evaluation - on one host core, run to completion; result as JSON next to the input."""
    import pickle
    import numpy as np   # noqa: F401
    with open(path, "rb") as f:
        job = pickle.load(f)
    from oracle.fem_oracle import lib, solve_batch
    lib()                                            # compile / load outside the timed span
    z, I = job["source"]
    ez = list(job["evals"])
    t0 = time.time()
    out, rc, st = solve_batch(job["mesh"], job["sigma"], [0, len(z)], list(z), list(I), [0, len(ez)], ez, condense=True, rtol=job["rtol"], maxit=job["maxit"])
    dt = time.time() - t0
    with open(path + ".json", "w") as f:
        json.dump(dict(seconds=dt, rc=int(rc), iterations=st["iterations"], n=st.get("n"), out=[float(v) for v in out]), f)


def cpu_baseline_start(work, rtol, maxit):
    """The oracle (scalar C port of the same algorithm) the way ONE worker of the reference's farm runs (remo3d.py:592-595,
    worker.py:104-112: a single-threaded process that assembles and solves a right-hand side on its own): started as a child
    process on one host core right after the meshes exist, so that its minutes of CPU time pass beside the GPU legs instead of
    behind them (at the reference's resolution one right-hand side is ~2 M unknowns and 750 Jacobi-PCG steps: 1.5-3 minutes of
    one core - the smallest complete sample there is).  Reported beside the GPU number, never the target."""
    import pickle
    import tempfile
    w = work[0]
    fd, path = tempfile.mkstemp(prefix="remo_cpu_leg_", suffix=".pkl")
    with os.fdopen(fd, "wb") as f:
        pickle.dump(dict(mesh=w["mesh"], sigma=w["sigma"], source=w["sources"][0], evals=w["evals"][0], rtol=rtol, maxit=maxit), f)
    env = dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-leg", path], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return dict(proc=proc, path=path, points=len(w["readers"][0]), t0=time.time())


def cpu_baseline_finish(leg, timeout_s=600.0):
    """Wait for the child of cpu_baseline_start.  Returns (record, potentials of batch 0 / RHS 0) or (record with an error, None)."""
    proc, path = leg["proc"], leg["path"]
    try:
        proc.wait(timeout=max(1.0, timeout_s))
    except subprocess.TimeoutExpired:
        proc.kill()
        return dict(value=None, unit="points/s", cores=1, kind="port", sample="the CPU leg did not finish within %.0f s" % timeout_s), None
    try:
        with open(path + ".json") as f:
            r = json.load(f)
    except OSError:
        return dict(value=None, unit="points/s", cores=1, kind="port", sample="the CPU leg failed: " + proc.stderr.read().decode()[-300:]), None
    finally:
        for q in (path, path + ".json"):
            try:
                os.remove(q)
            except OSError:
                pass
    import numpy as np
    pts = leg["points"]
    return dict(value=pts / r["seconds"], unit="points/s", cores=1, kind="port", seconds=r["seconds"], pcg_iterations=r["iterations"],
                sample=f"ONE right-hand side of the workload (batch 0, RHS 0: {pts} point(s), {r.get('n')} unknowns) on ONE host core, run to completion beside the GPU legs: "
                       f"oracle/fem_oracle.c assembly + Jacobi-PCG (rtol as the GPU run, {r['iterations']} steps, rc {r['rc']}) + evaluation in {r['seconds']:.1f} s.  "
                       "The reference farms such workers out one per core (remo3d.py:592-595); NGSolve is not installable here, so this is the build's "
                       "scalar C restatement, not the reference binary"), np.asarray(r["out"])


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


def pmc_traffic(workload, n_free, nnz, op="csr"):
    """HBM bytes per operator application from the committed rocprofv3 --pmc passes of this workload's mesh size and operator
    (profiles/, collected with tools/collect_traffic.sh + tools/pmc_traffic.py on the first 20 depths of the sweep: the profiled
    run meshes in-process); None when size or operator differ."""
    import re
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            p = json.load(f)
    except OSError:
        return None
    size = re.search(r"mesh size (\w+)", workload or "")
    if size and p.get("mesh_size") == size.group(1) and p.get("operator", "csr") == op:
        return dict(bytes=p["spmm"]["traffic_bytes_per_launch"], n_free=p.get("n_free"), mesh_T=p.get("mesh_T"))
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
    """The timed part: resident batches of one context set, one pass = one_step()."""

    def __init__(self, work, n_depths, local, opts, streams=1, schedule="static", all_resident=False):
        import numpy as np
        from remo3d_amd import solver, sweep, tasks
        self.np, self.sweep, self.tasks = np, sweep, tasks
        self.work, self.n_depths, self.opts, self.schedule = work, n_depths, opts, schedule
        self.ctxs = [solver.Context(local) for _ in range(max(1, streams))]
        # dynamic schedule (all_resident: the rank holds the host data of every batch, the queue index is the batch index): nothing
        # is uploaded ahead - a drawn batch is brought in by an uploader context on its own stream while the one before it runs
        self.all_resident = all_resident
        self.uploader = solver.Context(local) if all_resident else None
        self.resident = [] if all_resident else [self.ctxs[i % len(self.ctxs)].batch(w["mesh"], w["sigma"], w["sources"], w["evals"]) for i, w in enumerate(work)]
        self.pool = None
        if len(self.ctxs) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=len(self.ctxs))

    def close(self):
        for b in self.resident:
            b.close()
        for c in self.ctxs + ([self.uploader] if self.uploader else []):
            c.close()

    def _collect(self, i, rc, slab, agg, b=None):
        np = self.np
        w = self.work[i]
        b = self.resident[i] if b is None else b
        st = b.stats
        if rc < 0:
            for rd in w["readers"]:
                for (di, ti, K, o, m) in rd:
                    slab[di, ti] = np.nan
            return
        agg["not_converged"] += int(rc == 1)
        outs = b.fetch()
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

    def one_step(self, h2d_inclusive=False):
        np = self.np
        slab = np.zeros((self.n_depths, len(TOOLS)))
        agg = dict(spmv_ms=0.0, spmv_ms_raw=0.0, spmv_launches=0, spmv_bytes_total=0.0, pcg_steps=0, not_converged=0, ms_symbolic=0.0,
                   ms_assemble=0.0, ms_solve=0.0, ms_h2d=0.0, ms_eval=0.0, n=0, nnz=0, max_it=0, batches=0, ev_over=0.0, op_used=0)
        t_busy = time.time()
        if h2d_inclusive:     # the host-buffer entry: every batch is copied to the device inside the timed span (remo_solve_batch)
            for w in self.work:
                outs, st, rc = self.ctxs[0].solve_batch(w["mesh"], w["sigma"], w["sources"], w["evals"], self.opts, raise_on_error=False)
                for u, rd in zip(outs, w["readers"]):
                    for (di, ti, K, o, m) in rd:
                        slab[di, ti] = self.tasks.apparent_resistivity(u[o:o + m], m, K, w["mesh"].dim) if rc >= 0 else np.nan
                agg["ms_h2d"] += st["ms_h2d"]; agg["batches"] += 1
        elif self.pool is not None:   # one host thread per context, each walks its own batches in order (ctypes releases the GIL)
            def drive(j):
                return [(i, self.resident[i].run(self.opts, raise_on_error=False)) for i in range(j, len(self.resident), len(self.ctxs))]
            for i, rc in sorted(p for chunk in self.pool.map(drive, range(len(self.ctxs))) for p in chunk):
                self._collect(i, rc, slab, agg)
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
                    self._collect(cur_i, cur.run(self.opts, raise_on_error=False, ctx=self.ctxs[0]), slab, agg, b=cur)
                    self.last_fetch = cur.fetch() if cur_i == 0 else getattr(self, "last_fetch", None)
                    cur.close()
        else:
            for i in range(len(self.work)):
                self._collect(i, self.resident[i].run(self.opts, raise_on_error=False), slab, agg)
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


def roofline_of(agg, precision, stride, workload_name=None):
    """The operator application of the CG (what `time_kernels` brackets: every launch of it) against the HBM roofline, priced by
    ITS OWN algorithmic bytes (remo_stats_t.spmv_bytes): CSR product 12 nnz + 4 n + 16 k n; patch operator 16 k n + 88 T (x read and
    y written once, 40 bytes of local indices + 48 of metric terms per element - no stored entries)."""
    ach = (agg["spmv_bytes_total"] / 1e9) / (agg["spmv_ms"] / 1e3) if agg["spmv_ms"] > 0 else None
    op = {0: "csr", 1: "element", 3: "patch"}.get(int(agg.get("op_used", 0)), "csr")
    tr = pmc_traffic(workload_name, int(agg["n"]), int(agg["nnz"]), op) if (workload_name and precision == "fp64") else None
    r = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=(ach / HBM_PEAK_GBS) if ach else None,
             traffic=tr["bytes"] if tr else None)
    if tr:     # the counter passes ran on the first batches of the 20-depth sweep of the same mesh size (in-process meshing under the profiler)
        r["traffic_measured_on"] = dict(n_free=tr["n_free"], mesh_T=tr["mesh_T"],
                                        algorithmic_bytes_there=(16.0 * 5 * tr["n_free"] + 88.0 * tr["mesh_T"]) if op == "patch" and tr["n_free"] and tr["mesh_T"] else None)
    prec = "fp64" if precision == "fp64" else "fp32 values and vectors"
    kernel = {"csr": "k_spmm_pair (CSR SpMM, %s, k=5 interleaved RHS)",
              "element": "k_elem_apply + k_elem_reduce (element-wise operator with a slab of element results, %s; `achieved` prices the CSR product's bytes)",
              "patch": "k_patch_apply (matrix-free patch operator, %s, k=5 interleaved RHS; inside the PCG the rows shared by several patches are summed by the update launch and <p, A p> is added by the patches themselves: one launch per application)"}[op] % prec
    fp = precision == "fp64"
    formula = {"csr": "12*nnz + 4*n + 16*k*n (SURVEY.md 8d)" if fp else "8*nnz + 4*n + 8*k*n (SURVEY.md 8d, fp32 storage)",
               "element": "12*nnz + 4*n + 16*k*n (the CSR product's figure)" if fp else "8*nnz + 4*n + 8*k*n",
               "patch": "16*k*n + 88*T (x and y once, 40 B local indices + 48 B metric terms per element)" if fp else "8*k*n + 88*T"}[op]
    r.update(kernel=kernel, operator=op,
             timed="every %d-th application of every solve, HIP events on the solver's stream, over the timed steps" % stride,
             launches=int(agg["spmv_launches"]),
             avg_launch_us=(1e3 * agg["spmv_ms"] / agg["spmv_launches"]) if agg["spmv_launches"] else None,
             avg_bracket_us_raw=(1e3 * agg["spmv_ms_raw"] / agg["spmv_launches"]) if agg["spmv_launches"] else None,
             empty_event_pair_us=1e3 * agg["ev_over"],
             bytes_per_launch=formula,
             algorithmic_bytes_per_launch=(agg["spmv_bytes_total"] / agg["spmv_launches"]) if agg["spmv_launches"] else None,
             traffic_unit="bytes per application: reads sized by the TCC_EA0_RDREQ 32/64/128-B request counters + WRITE_SIZE over the kernels of the bracket, " + PMC_FILE)
    # the compute side of the same launches: a matrix-free operator trades bytes for arithmetic, and above ~10 flop per byte
    # (78.6 TFLOP/s over 8 TB/s) the fp64 rate is the roof that binds, not HBM
    fl = agg.get("spmv_flops_total", 0.0)
    if fl > 0 and agg["spmv_ms"] > 0 and precision == "fp64":
        tf = fl / (agg["spmv_ms"] / 1e3) / 1e12
        r["compute"] = dict(flops_per_launch=fl / agg["spmv_launches"], achieved_tflops=tf, peak_tflops=FP64_PEAK_TFLOPS, frac=tf / FP64_PEAK_TFLOPS,
                            flop_per_algorithmic_byte=fl / agg["spmv_bytes_total"], machine_balance_flop_per_byte=FP64_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS,
                            note="fp64 vector (= matrix) rate of MI355X; flops counted from the kernel's ISA (patch operator) or 2 nnz k (CSR product)")
    return r


def log(msg):
    """Progress on stderr (stdout carries the ONE JSON line)."""
    if int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("[bench %7.1fs] %s\n" % (time.time() - T_START, msg))
        sys.stderr.flush()


T_START = time.time()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-leg":
        return _cpu_leg_main(sys.argv[2])
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
    ap.add_argument("--op", default="auto", choices=["auto", "csr", "element", "patch"],
                    help="how the CG applies A: csr = SpMM on the assembled matrix; element = element-wise operator through the factorised "
                         "reference tensors (slab of element results); patch = the same tensors patch by patch with LDS-staged vectors (round 3); "
                         "auto (the library's default) = patch in 3D")
    ap.add_argument("--streams", type=int, default=1,
                    help="contexts (HIP streams + arenas) per GPU, each driven by its own host thread over its share of the batches; "
                         "1 = the headline configuration (per-launch SpMM timing is only meaningful without overlap)")
    ap.add_argument("--overlap", default="prepare", choices=["prepare", "all"],
                    help="with --streams > 1: 'prepare' = only one batch is in its PCG at a time, the other contexts number / assemble "
                         "theirs beside it; 'all' = no restriction")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--sizes", default="L/csr:4,L/mixed:4,S:8,S/csr:8,M:8,XL:2,XL/mixed:2,conforming-M:8,2D-BM1:8",
                    help="further workloads measured in the same run at N = 1 (SIZE:batches, 'conforming-' prefix = conforming meshes, "
                         "'/mixed' = fp32 PCG in fp64 refinement, '/element' / '/csr' = that operator instead of the choice by size; '2D-BM1' = BASELINE configs[1], "
                         "Benchmark model 1 in 2D, '/chebyshev' = polynomial instead of the multigrid cycle on the vertex block, '/2ctx' = two contexts (streams, host threads) share the batches), reported in the `sizes` array; '' = none")
    ap.add_argument("--no-extras", action="store_true", help="skip the `sizes` and H2D-inclusive legs")
    ap.add_argument("--mesh-workers", type=int, default=10, help="CPU processes that build the synthetic meshes side by side (before any GPU work)")
    ap.add_argument("--coarse", default="", metavar="DEGREE,RATIO", help="experiments only: Chebyshev degree and interval ratio of the P1 block (default: by vertex count)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B experiments only: remo_debug_tune(KEY, VALUE) before the run (include/remo3d_hip_debug.h lists the keys)")
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
    extras = (world == 1) and not args.no_extras and args.streams == 1 and args.precision == "fp64" and args.mesh == "lattice" and not args.tune and args.op == "auto" and not args.coarse
    extra_specs = []
    if extras and args.sizes:
        for spec in args.sizes.split(","):
            name, nb = spec.split(":")
            conf = name.startswith("conforming-")
            size = name.split("/")[0].split("-")[-1]
            if name.startswith("2D"):
                extra_specs.append((name, "2D", "2D", int(nb)))
            elif not (name == args.size):
                extra_specs.append((name, size, "conforming" if conf else "lattice", int(nb)))

    # ---- synthetic meshes: CPU-only worker processes, before this process loads the HIP library ----
    mesh_pool = None
    profiled = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_LIBRARY", "HSA_TOOLS_LIB"))
    if profiled:
        log("running under a profiler: meshes are built in this process (no child processes behind a preloaded tool library)")
    # every rank builds the meshes of its own batches in its own CPU-only worker processes (N ranks share the node's cores)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 8
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
                                         build_workload(0, 1, 20, SIZES[size], mesh_3d=kind, max_batches=nb, pool=mesh_pool, wait=False))
    wl = build_workload(rank, world, args.depths, SIZES[args.size], mesh_3d=args.mesh, total_depths=args.total_depths or None,
                        all_batches=dynamic, pool=mesh_pool, wait=False)

    def progress(what):
        return lambda done, total: log("meshes: %s %d / %d" % (what, done, total)) if (done % 5 == 0 or done == total) else None
    all_futs = [f for pend in list(pending.values()) + [wl] if isinstance(pend, Pending) for f in pend.futs]
    import threading
    stop_beat = threading.Event()

    def beat():       # the mesh phase of the reference-resolution workload takes minutes: say so while it lasts
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

    cpu_leg = None
    if not args.no_cpu and world == 1 and not profiled:           # the CPU leg belongs to the N = 1 line only; it runs beside the GPU legs
        cpu_leg = cpu_baseline_start(wl["work"], args.rtol, 20000)
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
    opts = solver.make_opts(preconditioner="multigrid", condense=True, rtol=args.rtol, maxsteps=args.maxsteps,
                            time_kernels=0 if args.no_events else stride, precision=args.precision,
                            serialize_solves=(args.streams > 1 and args.overlap == "prepare"), op=args.op,
                            coarse_degree=int(args.coarse.split(",")[0]) if args.coarse else 0, coarse_ratio=int(args.coarse.split(",")[1]) if args.coarse else 0)
    work = wl["work"]
    n_depths = len(wl["depths"])
    runner = Runner(work, n_depths, local, opts, streams=args.streams, schedule=args.schedule, all_resident=dynamic)

    def sync():
        if dist_on:
            import torch
            sweep.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    log("batches resident on the device; timing %d + %d steps" % (args.warmup, args.steps))
    dt_local, slab, agg, busy = timed(runner, args.steps, args.warmup, sync)
    log("timed region done: %.3f s" % dt_local)
    dt = sweep.max_over_ranks(dt_local)
    busy_ranks = sweep.gather_floats([1e3 * busy / args.steps, float(agg["batches"])])

    n_points = n_depths * len(TOOLS)
    value = n_points * args.steps / dt
    per = f"{args.total_depths} depths in all" if strong else f"{args.depths} depths/GPU"
    workload_name = f"BM3 dip30, tools A0.4M6.0N+A2.0M0.5N, {per}, R=50, batch 5, mesh size {args.size}"
    if args.mesh != "lattice":
        workload_name += ", interface-conforming revolved meshes"
    if rank != 0:
        runner.close()
        return
    out = dict(metric="measurement points/sec (3D benchmark model)", value=value, unit="points/s", n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="strong" if strong else "weak",
               vs_baseline=None, dtype="f64" if args.precision == "fp64" else "f32 PCG inside f64 residual refinement", data="synthetic",
               config=dict(workload=workload_name, schedule=args.schedule if world > 1 else "single rank",
                           batches_total=wl["n_batches"], batches_rank0=int(agg["batches"]), rhs_rank0=sum(len(w["sources"]) for w in work) if not dynamic else None,
                           points_total=n_points, mesh_T=int(work[0]["mesh"].n_elems), n_free=int(agg["n"]), nnz=int(agg["nnz"]), rtol=args.rtol,
                           maxsteps=args.maxsteps, precision=args.precision, streams_per_gpu=args.streams, operator={0: "csr", 1: "element", 3: "patch"}.get(int(agg["op_used"]), "csr"),
                           preconditioner="multigrid = Chebyshev polynomial on the P1 vertex block (degree / interval by vertex count: 5 on lmax/90..lmax at 12.6 k vertices, 12 on lmax/320 at 83 k) + Jacobi on edge/face dofs",
                           max_pcg_iterations=int(agg["max_it"]), batches_not_converged=int(agg["not_converged"]), nan_points=int(np.isnan(slab).sum())),
               roofline=roofline_of(agg, args.precision, stride, workload_name),
               breakdown_ms_per_step=dict(numbering_and_pattern_device=agg["ms_symbolic"], h2d_points=agg["ms_h2d"], assemble=agg["ms_assemble"], solve=agg["ms_solve"],
                                          eval=agg["ms_eval"], pcg_steps=int(agg["pcg_steps"]), mesh_generation_excluded_s=wl["mesh_s"],
                                          all_mesh_generation_wall_s=mesh_wall),
               per_rank=dict(busy_ms_per_step=[b[0] for b in busy_ranks], batches_last_step=[int(b[1]) for b in busy_ranks]))
    if args.tune:
        out["config"]["debug_tune"] = list(args.tune)
    # assembly against ITS roofline (8 nnz values written once + 4 nnz column reads + T (4 (dim + 1) + 4) + 8 dim nv bytes)
    if agg["batches"] and agg["ms_assemble"] > 0:
        m0 = work[0]["mesh"]
        asm_bytes = 12.0 * agg["nnz"] + m0.n_elems * (4 * 4 + 4) + 8.0 * 3 * m0.n_nodes
        us = 1e3 * agg["ms_assemble"] / agg["batches"]
        out["assembly"] = dict(kernels="k_metric_terms + k_assemble (HIP events around both)", algorithmic_bytes_per_batch=asm_bytes, us_per_batch=us,
                               achieved_GBs=asm_bytes / us / 1e3, frac_of_hbm_peak=asm_bytes / us / 1e3 / HBM_PEAK_GBS)

    if extras:
        # the same sweep with the per-batch host -> device copy of the mesh arrays INSIDE the timed span (SURVEY 8d's span; the
        # one-shot entry remo_solve_batch: create + run + fetch + destroy per batch)
        dth, slab_h, agg_h, _ = timed(runner, max(1, min(args.steps, 2)), 1, sync, h2d_inclusive=True)
        log("H2D-inclusive leg done: %.3f s" % dth)
        # two contexts on the GPU (HIP streams + arenas, one host thread each, batches dealt alternately): what Model does with
        # gpu_workers = 2.  The launch-latency-bound quarter of one batch's PCG step is filled by the other batch's kernels; the
        # per-kernel figures above are NOT taken from this leg (kernels that share the chip are not timed in isolation).
        r_two = Runner(work, n_depths, local, solver.make_opts(preconditioner="multigrid", condense=True, rtol=args.rtol, maxsteps=args.maxsteps),
                       streams=2)
        dt2c, slab_2c, _, _ = timed(r_two, max(1, min(args.steps, 2)), 1, sync)
        r_two.close()
        out["value_two_contexts"] = dict(value=n_points * max(1, min(args.steps, 2)) / dt2c, unit="points/s",
                                         max_abs_log_diff_vs_one_context=float(np.nanmax(np.abs(slab_2c - slab))))
        log("two-context leg done: %.3f s" % dt2c)
        out["value_h2d_inclusive"] = dict(value=n_points * max(1, min(args.steps, 2)) / dth, unit="points/s",
                                          note="remo_solve_batch per batch: upload of the mesh arrays (pageable host memory) + run + fetch inside the timed span",
                                          max_abs_log_diff_vs_resident=float(np.nanmax(np.abs(slab_h - slab))))
    got0 = runner.resident[0].fetch()[0].copy() if runner.resident else None
    runner.close()
    if extras:
        out["box"] = box_stream(local)

    sizes = []
    for name, w2 in extra_wl:
        prec2 = "mixed" if "/mixed" in name else args.precision
        op2 = "element" if "/element" in name else ("csr" if "/csr" in name else ("patch" if "/patch" in name else args.op))
        coarse2 = "chebyshev" if "/chebyshev" in name else "auto"
        opts2 = solver.make_opts(preconditioner="multigrid", condense=True, rtol=args.rtol, maxsteps=args.maxsteps,
                                 time_kernels=0 if args.no_events else stride, precision=prec2, op=op2, coarse=coarse2)
        r2 = Runner(w2["work"], len(w2["depths"]), local, opts2, streams=2 if "/2ctx" in name else 1)
        st2 = 2
        dt2, slab2, agg2, _ = timed(r2, st2, 1, sync)
        pts = sum(len(rd) for w in w2["work"] for rd in w["readers"])
        rf = roofline_of(agg2, prec2, stride)
        op2 = {0: "csr", 1: "element", 3: "patch"}.get(int(agg2["op_used"]), "csr")
        sizes.append(dict(workload=name, precision=prec2, operator=op2, vertex_block_solver={0: "none", 1: "chebyshev", 2: "multigrid cycle"}[agg2.get("coarse_used", 1)], contexts=2 if "/2ctx" in name else 1, batches=len(w2["work"]), points=pts, value=pts * st2 / dt2, unit="points/s", mesh_T=int(w2["work"][0]["mesh"].n_elems),
                          n_free=int(agg2["n"]), nnz=int(agg2["nnz"]), pcg_steps_per_batch=agg2["pcg_steps"] / max(1, agg2["batches"]),
                          max_pcg_iterations=int(agg2["max_it"]), operator_frac_of_hbm_peak=rf["frac"], operator_bytes=rf["bytes_per_launch"], apply_avg_launch_us=rf["avg_launch_us"],
                          solve_ms_per_batch=agg2["ms_solve"] / max(1, agg2["batches"]), nan_points=int(np.isnan(slab2).sum())))
        r2.close()
        log("size leg %s done: %.3f s for %d steps" % (name, dt2, st2))
    if extras:
        out["sizes"] = sizes
    out["cpu_baseline"] = None
    if cpu_leg is not None:
        cb, ref_out = cpu_baseline_finish(cpu_leg)
        log("CPU baseline leg done")
        out["cpu_baseline"] = cb
        if ref_out is not None and got0 is not None:
            out["config"]["gpu_vs_oracle_max_rel_diff_batch0_rhs0"] = float(np.max(np.abs(got0 - ref_out) / np.abs(ref_out)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
