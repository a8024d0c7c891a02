"""`Model`: the public face of ReMo3D (remo3d/remo3d.py class Model, lines 23-1147) on top of the
MI355X library.  Same constructor, same `compute_synthetic_logs(...)` keyword arguments and
defaults, same `logs` dictionary (tool -> [n_depths, 2] array of depth and apparent resistivity),
same `Results_<n>.txt` layout.

What differs, by design:
  * the MPI worker farm (remo3d.py:552-599, 809-865) is gone: one process drives one GPU; under
    `torchrun` every rank takes a block-cyclic share of the batches and the logs are combined
    with one RCCL all-reduce of the [n_depths, n_tools] slab (sweep.py);
  * Gmsh / Netgen are not available: batch meshes come from `mesh_provider` (default: the seeded
    in-repo mesher, meshgen.make_mesh, with the reference's size field); a provider that reads
    real MSH 2.2 files can be plugged in;
  * there is no CPU solver behind it: without libremo3d_hip.so and a GPU, construction of the
    solver context raises.
"""
from __future__ import annotations

import datetime
import os
import time
from typing import Callable, Dict, Optional

import numpy as np

from . import geometry, meshgen, tasks, tools as tools_mod

CONVERSION = {"M": 1.0, "DM": 0.1, "CM": 0.01, "MM": 0.001, "IN": 0.0254, "FT": 0.3048}


# Default multiplier on the reference's size field.  2D: 0.35 - the value at which the complete Example_01 of the reference
# (1506 points) sits within p99 8e-4 / max 2.2e-3 of the reference's committed log (scale 1.0: 7.7e-3 / 2.5e-2, 0.5: 3.6e-3 /
# 9.4e-3; profiles/r02_example01_scales.log): short lateral spacings read potential differences of a few per cent of the
# potential, next to a borehole wall with a kink every 0.1 m.  3D: 1.0 (the reference's field as it is).
DEFAULT_SCALE = {2: 0.35, 3: 1.0}

# Contexts (HIP stream + arena + host thread) per GPU when the caller does not say (gpu_workers = 0).  bench.py, size L, same box,
# points/s on SURVEY 8d's span: 1 context 91-101, 2: 109.7-110.3 / 115.3-115.9, 3: 113.3-113.4 / 117.2, 4: 110.1
# (profiles/r04_i_bench_streams*.json, r04_c_bench_streams*.json) - all with the HIP runtime's default of four hardware queues,
# where the fourth context shares a queue with another stream of the process.  With eight (remo3d_amd/__init__.py), end of round 4,
# alternating runs on one box: 3 contexts 159.9-160.0, 4: 157.9-160.6, 5: 168.1-168.3, 6: 165.7-166.5, 8: 163.8-165.2
# (profiles/r04_bl_hw_queues_and_contexts.json).
DEFAULT_CONTEXTS = 5
MAX_CONTEXTS = 8


def lattice_mesh_key(dim, domain_radius, batch, scale, seed=0) -> tuple:
    """What a lattice mesh depends on: the electrode pattern of the batch in its own frame, size multiplier and seed - not the
    depth (the 40 batches of the bench's 100-depth sweep share six meshes).  Key of the provider's caches, in memory and on disk."""
    cur = batch.electrodes[0, batch.electrodes[1, :] != 0]
    pot = batch.electrodes[0, batch.electrodes[1, :] == 0]
    return (int(dim), float(domain_radius), tuple(float(v) for v in np.round(cur, 4)), tuple(float(v) for v in np.round(pot, 4)), float(scale), int(seed))


def tuned_coarse_for_conforming(n_nodes: int) -> dict:
    """Solver of the P1 block for the revolved conforming 3D meshes.  Graded and sheared, they want more of it than isotropic meshes
    of the same vertex count: one multigrid cycle where its hierarchy can be built (GPU scan at 64 k vertices, tools/scan_conforming.py,
    profiles/r04_p_scan_conforming_vertex_solver.log: 62.4 ms of solve per batch and 187 steps against 69.5 ms / 198 steps with the best
    polynomial; Model end to end 63 against 54 points/s), else a Chebyshev polynomial of higher degree on a wider interval than the
    library's default (tools/scan_coarse3d.py: 8 / 300 at 19 k vertices, 14 / 400 at 51 k)."""
    rel = max(int(n_nodes), 1) / 12600.0
    return dict(coarse="amg_or_chebyshev", coarse_degree=int(min(16, max(6, round(7.0 * rel ** 0.5)))),
                coarse_ratio=int(min(1200, max(150, round(220.0 * rel ** (2.0 / 3.0))))))


def vertex_solver_options(dim: int, n_nodes: int, conforming: bool, n_contexts: int) -> dict:
    """The solver of the P1 block `Model` asks for when the caller did not choose one (keywords of solver.make_opts).
    2D: the library's default (one multigrid cycle).  3D, interface-conforming meshes of the default provider: the cycle with the tuned
    polynomial behind it (tuned_coarse_for_conforming).  3D, any other mesh: the cycle (polynomial if its hierarchy cannot be built)
    WHEN SEVERAL CONTEXTS SHARE THE GPU - its set-up waits for the host once per level and its twelve launches per step are latency,
    which other batches' kernels fill: bench.py on three contexts, points/s with the cycle against the polynomial: size S 434 / 410,
    M 253 / 222, L 131 / 122, L mixed 187 / 177, XL 37.0 / 33.9, XL mixed 54.6 / 49.3 (profiles/r04_s_*, r04_u_*, r04_x_*, r04_y_*); on ONE context the polynomial (the library's default in 3D)
    is ahead: L 93 / 87, S 252 / 205 (profiles/r04_t_*)."""
    if dim != 3:
        return {}
    if conforming:
        return tuned_coarse_for_conforming(n_nodes)
    return dict(coarse="amg_or_chebyshev") if n_contexts >= 2 else {}


def default_mesh_provider(scale: Optional[float] = None, seed: int = 0, mesh_3d: str = "conforming", sectors: int = 6) -> Callable:
    """Batch mesh factory.  scale: multiplier on the reference's size field (None: DEFAULT_SCALE by dimension).  2D: interface-conforming half-disc meshes built per batch.
    3D, mesh_3d = "conforming" (default): the 2D conforming mesh of the window revolved in the sheared
    frame of the dipping layers (meshgen.make_mesh_3d_conforming: every interface of the reference's
    OpenCASCADE geometry is a union of element faces); "lattice": seeded graded half-ball meshes cached on
    the electrode pattern, materials by element centroid (the bench's synthetic meshes)."""
    if mesh_3d not in ("conforming", "lattice"):
        raise ValueError("mesh_3d must be 'conforming' or 'lattice'")
    cache: Dict[tuple, meshgen.Mesh] = {}

    scale_arg = scale

    def provider(dim, domain_radius, batch, local_formation_geometry, local_borehole_geometry, dip_rad):
        scale = DEFAULT_SCALE[dim] if scale_arg is None else scale_arg
        cur = batch.electrodes[0, batch.electrodes[1, :] != 0]
        pot = batch.electrodes[0, batch.electrodes[1, :] == 0]
        fn = meshgen.layered_material_fn(dim, local_formation_geometry, local_borehole_geometry, dip_rad)
        if dim == 2:
            # axisymmetric models get meshes that CONFORM to the borehole wall, the layer boundaries and
            # the flushed-zone radii (like the reference's Netgen / Gmsh geometry); they depend on the
            # depth window, so there is nothing to cache
            # and are refined around measuring electrodes as well as current electrodes: a potential
            # difference read next to a layer boundary needs it (max deviation from the reference's
            # Example_01 log 7e-2 -> 5e-3, median 9e-4 -> 3e-4; it costs 2.5x the triangles, which is cheap in 2D)
            polys = meshgen.layer_interfaces_2d(local_formation_geometry, local_borehole_geometry, domain_radius)
            inside = [z for z in list(cur) + list(pot) if abs(z) < domain_radius]
            fg = np.asarray(local_formation_geometry, dtype=float)
            cap = meshgen.LayerCap(np.concatenate([fg[:1, 0], fg[:, 1]]))     # thin beds bound the element size
            return meshgen.make_mesh(2, domain_radius, sources_z=inside, scale=scale, seed=seed, interfaces=polys, material_fn=fn,
                                     layer_cap=cap)
        if mesh_3d == "conforming":
            fg = np.asarray(local_formation_geometry, dtype=float)
            cap = meshgen.LayerCap(np.concatenate([fg[:1, 0], fg[:, 1]]))
            return meshgen.make_mesh_3d_conforming(domain_radius, local_formation_geometry, local_borehole_geometry, dip_rad,
                                                   sources_z=list(cur), snap_z=list(pot), scale=scale, seed=seed, layer_cap=cap, sectors=sectors)
        key = lattice_mesh_key(dim, domain_radius, batch, scale, seed)
        base = cache.get(key)
        if base is None:
            # (also kept on disk, meshgen.cached_mesh: every batch of a tool, every rank and every rerun share it)
            base = meshgen.cached_mesh(("lattice",) + key, lambda: meshgen.make_mesh(
                dim, domain_radius, sources_z=list(cur), scale=scale, seed=seed, snap_z=[z for z in pot if abs(z) < domain_radius]))
            if len(cache) > 64:
                cache.clear()
            cache[key] = base
        mat = fn(base.coords[base.conn].mean(1)).astype(np.int32)
        return meshgen.Mesh(dim, base.coords, base.conn, mat, base.bconn, base.bdirichlet, base.meta)

    return provider


# mesh generation ahead of the solver, in spawned processes (numpy / scipy only: they never touch the GPU)
_WORKER_PROVIDER = None


def _mesh_worker_init(provider_kwargs):
    global _WORKER_PROVIDER
    _WORKER_PROVIDER = default_mesh_provider(**provider_kwargs)


def _mesh_worker_run(job):
    import types
    dim, domain_radius, electrodes, fg, bh, dip_rad = job
    return _WORKER_PROVIDER(dim, domain_radius, types.SimpleNamespace(electrodes=electrodes), fg, bh, dip_rad)


class Model:
    conversion_table = CONVERSION

    def __init__(self, tools, force_single_electrode_configuration=True):
        self.tools, self.sec = self.set_tools_parameters(tools, force_single_electrode_configuration=force_single_electrode_configuration)
        self.formation_model = None
        self.borehole_model = None
        self.dip_deg = None
        self.dip_rad = None
        self.cpu_workers = None
        self.gpu_workers = None
        self.ctx = None
        self.extra_ctx = []
        self.logs = None
        self.timing = {}

    # -- complete procedure (remo3d.py:65-174) ---------------------------------------------------
    @classmethod
    def compute_synthetic_logs(cls, tools, measurement_depths, formation_model, borehole_model,
                               force_single_electrode_configuration=True, formation_units=["M", "M", "M"],
                               borehole_geometry_type="diameter", borehole_units=["M", "M"], dip=0, cpu_workers=4,
                               gpu_workers=0, domain_radius=50, batch_size=5, mesh_generator="auto",
                               preconditioner="multigrid", condense=True, **extensions):
        model = cls(tools, force_single_electrode_configuration=force_single_electrode_configuration)
        model.set_model_parameters(formation_model, borehole_model, borehole_geometry_type=borehole_geometry_type, dip=dip)
        model.initialize_workers(cpu_workers=cpu_workers, gpu_workers=gpu_workers)
        model.simulate_logs(measurement_depths, domain_radius=domain_radius, batch_size=batch_size, mesh_generator=mesh_generator,
                            preconditioner=preconditioner, condense=condense, **extensions)
        model.shutdown_workers()
        return model

    # -- tools (remo3d.py:178-321) ---------------------------------------------------------------
    def set_tools_parameters(self, tools, force_single_electrode_configuration=True):
        return tools_mod.tool_tables(tools, force_single_electrode_configuration)

    # -- model (remo3d.py:344-548) ---------------------------------------------------------------
    def set_model_parameters(self, formation_model, borehole_model, borehole_geometry_type="diameter", dip=0):
        """Formation and borehole model from table files (paths) or from arrays in metres (remo3d.py:344-377: same argument
        meaning; an argument of any other type is ignored there and is a TypeError here)."""
        def model_from(source, from_file, from_array, what):
            if isinstance(source, (str, os.PathLike)):
                return from_file(os.fspath(source))
            if isinstance(source, np.ndarray):
                return from_array(source)
            raise TypeError("{} model has to be a file name or a numpy array, not {}".format(what, type(source).__name__))
        self.formation_model = model_from(formation_model, self.load_formation_parameters, self.set_formation_parameters, "formation")
        self.borehole_model = model_from(borehole_model, lambda f: self.load_borehole_parameters(f, borehole_geometry_type),
                                         lambda a: self.set_borehole_parameters(a, borehole_geometry_type), "borehole")
        self.dip_deg, self.dip_rad = self.set_dip(dip)
        self._check_model_geometry()

    @staticmethod
    def _read_table(path):
        """Tab-separated table with a name row and a unit row (remo3d.py:395-398, 459-462)."""
        with open(path) as f:
            lines = f.read().splitlines()
        units = lines[1].split()
        rows = [ln.split("\t") for ln in lines[2:] if ln.strip()]
        data = np.array([[float(v) for v in r if v.strip() != ""] for r in rows], dtype=float)
        return np.atleast_2d(data), units

    def load_formation_parameters(self, formation_model_file):
        data, units = self._read_table(formation_model_file)
        return self.set_formation_parameters(data, units[:-2])

    def set_formation_parameters(self, formation_parameters, formation_units=["M", "M", "M"]):
        fp = np.array(formation_parameters, dtype=float)   # a copy (the reference converts the caller's array in place, remo3d.py:427)
        for i, u in enumerate(formation_units):
            if u not in CONVERSION:
                raise ValueError("{} unit in formation model file not recognized. Allowed units: M, DM, CM, MM, IN, FT".format(u))
            fp[:, i] *= CONVERSION[u]
        if (np.diff(fp[:, :2], axis=0) <= 0.0).any() or (fp[1:, 0] != fp[:-1, 1]).any():
            raise ValueError("Uncorrect formation model geometry")
        if np.nanmin(fp[:, [3, 4]]) <= 0.0:
            raise ValueError("Formation resistivies have to be higher than 0 ohmm")
        return fp

    def load_borehole_parameters(self, borehole_model_file, borehole_geometry_type="diameter"):
        data, units = self._read_table(borehole_model_file)
        return self.set_borehole_parameters(data, borehole_geometry_type=borehole_geometry_type, borehole_units=units[:-1])

    def set_borehole_parameters(self, borehole_parameters, borehole_geometry_type="diameter", borehole_units=["M", "M"]):
        bp = np.array(borehole_parameters, dtype=float)   # a copy: the caller's array is not converted in place
        if np.shape(bp)[0] < 2:
            raise ValueError("Borehole paramaters have to be defined for at least two depths")
        for i, u in enumerate(borehole_units):
            if u not in CONVERSION:
                raise ValueError("{} unit in borehole model file not recognized. Allowed units: M, DM, CM, MM, IN, FT".format(u))
            bp[:, i] *= CONVERSION[u]
        if (np.diff(bp[:, 0], axis=0) <= 0.0).any() or (bp[:, 1] <= 0.0).any():
            raise ValueError("Uncorrect borehole model geometry")
        if borehole_geometry_type == "diameter":
            bp[:, 1] /= 2
        elif borehole_geometry_type != "radius":
            raise ValueError("Uncorrect borehole geometry type - use 'diameter' or 'radius' to specify borehole geometry")
        if np.nanmin(bp[:, 2]) <= 0.0:
            raise ValueError("Drilling mud resistivies have to be higher than 0 ohmm")
        return bp

    def set_dip(self, dip):
        if dip < 0 or dip >= 90:
            raise ValueError("Uncorrect dip angle")
        return dip, dip * np.pi / 180

    def _check_model_geometry(self):
        for i in range(np.shape(self.formation_model)[0]):
            inside = (self.borehole_model[:, 0] >= self.formation_model[i, 0]) & (self.borehole_model[:, 0] <= self.formation_model[i, 1])
            if np.any(self.borehole_model[inside, 1] >= self.formation_model[i, 2]):
                raise ValueError("Borehole radius have to be smaller than the extend of the filtration zone")

    def _add_points_to_borehole(self, maximal_distance=0.15):
        """Densify the borehole polyline for 3D models (remo3d.py:694-720).  Unlike the reference
        (Appendix A of SURVEY.md: unbound variable) an already dense model is returned unchanged."""
        bm = self.borehole_model
        depths = [bm[0, 0]]
        for i in range(1, bm.shape[0]):
            gap = bm[i, 0] - bm[i - 1, 0]
            if gap > maximal_distance:
                depths += list(np.linspace(bm[i - 1, 0], bm[i, 0], np.max([3, int(gap * 10 + 1)]))[1:])
            else:
                depths.append(bm[i, 0])
        depths = np.asarray(depths)
        if depths.shape[0] > bm.shape[0]:
            return np.vstack([depths, np.interp(depths, bm[:, 0], bm[:, 1]), np.interp(depths, bm[:, 0], bm[:, 2])]).T
        return bm

    # -- workers (remo3d.py:552-599, 887-899) ------------------------------------------------------
    def initialize_workers(self, cpu_workers=4, gpu_workers=0, context_factory: Optional[Callable] = None):
        """The reference spawns MPI workers here (remo3d.py:552-599); this build opens GPU contexts in the
        calling process (device = LOCAL_RANK under torchrun): `gpu_workers` of them (0, the default, = DEFAULT_CONTEXTS), each
        with its own HIP stream and arena and driven by its own host thread in simulate_logs, so batches
        overlap on the GPU the way the reference's GPU workers overlap (they fill the launch-latency gaps of
        one another: +40 % in 3D at five, more in 2D).  Parallelism ACROSS GPUs comes from the launcher (one rank per GPU);
        cpu_workers is validated like the reference and sizes the pool of mesh-generating processes (there is no CPU
        solver).  Under torchrun this is also where the rank joins the process group (sweep.init_from_env)."""
        if type(cpu_workers) != int or type(gpu_workers) != int:
            raise ValueError("The number of processes have to be an intager")
        if cpu_workers < 1:
            raise ValueError("Minimal number of cpu workers is 1")
        if gpu_workers < 0:
            raise ValueError("Minimal number of gpu workers is 0")
        self.cpu_workers, self.gpu_workers = cpu_workers, gpu_workers
        from . import solver, sweep
        # under torchrun (WORLD_SIZE > 1) the ranks share the sweep: join the process group here, as the reference
        # brings its farm up in initialize_workers (remo3d.py:592-599); without it every rank would compute every batch
        sweep.init_from_env()
        self.world_size = sweep.world_size()
        device = int(os.environ.get("REMO_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        make = context_factory or solver.Context     # context_factory(device): a stand-in solver for the CPU tests of the sweep
        self.ctx = make(device)
        # contexts on this GPU: gpu_workers of them (at most MAX_CONTEXTS); the reference's default gpu_workers = 0 means "no GPU worker"
        # there and "the build's default" here: DEFAULT_CONTEXTS - the launch-latency-bound quarter of one batch's PCG step (the
        # chain of small launches on the vertex block) is filled by the other batches' kernels
        n_ctx = DEFAULT_CONTEXTS if gpu_workers == 0 else min(gpu_workers, MAX_CONTEXTS)
        self.extra_ctx = [make(device) for _ in range(n_ctx - 1)]

    def shutdown_workers(self):
        for c in getattr(self, "extra_ctx", []):
            c.close()
        self.extra_ctx = []
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None

    # -- tasks (remo3d.py:602-692) -----------------------------------------------------------------
    def _prepare_simulation_depths_and_tasks(self, measurement_depths, batch_size):
        return tasks.build_batches(self.tools, self.sec, measurement_depths, batch_size)

    # -- the sweep (remo3d.py:723-884 + workers/worker.py:74-142) ----------------------------------
    def simulate_logs(self, measurement_depths, domain_radius=50, batch_size=5, mesh_generator="auto", preconditioner="multigrid",
                      condense=True, mesh_provider: Optional[Callable] = None, mesh_scale: Optional[float] = None, rtol: float = 1e-8,
                      maxsteps: int = 1000, verbose: bool = True, mesh_workers: Optional[int] = None, precision: str = "fp64",
                      schedule: str = "static", solver_options: Optional[dict] = None):
        """solver_options: further keywords of solver.make_opts for every batch (op, coarse, quadrature, assemble, ...)."""
        from . import solver, sweep
        extra = dict(solver_options or {})
        start = time.time()
        measurement_depths = np.asarray(measurement_depths, dtype=float)
        alert = False
        for t in self.tools.values():
            far = np.max(np.abs(t[0, :3]))
            if far > domain_radius:
                raise ValueError("Some electrodes are locate outside the simulation domain. Domain size have to be increased")
            alert |= far > 0.75 * domain_radius
        if alert and verbose:
            print("Some electrodes are located close to the boundary of the simulation domain. This may cause problems during simulation. "
                  "Consider increase of the domain size")
        if mesh_generator not in ("auto", "gmsh", "netgen"):
            raise ValueError("mesh_generator has to be 'auto', 'gmsh' or 'netgen'")
        is3d = not np.isclose(self.dip_deg, 0)
        if is3d and mesh_generator == "netgen":
            raise ValueError("The only mesh generator supported in 3D models is gmsh")
        if preconditioner not in ("local", "multigrid"):
            raise ValueError("preconditioner has to be 'local' or 'multigrid'")
        if self.ctx is None:
            raise RuntimeError("initialize_workers() has to be called before simulate_logs()")
        if self.dip_deg != 0:
            self.borehole_model = self._add_points_to_borehole()
        dim = 3 if is3d else 2
        netgen_path = (not is3d) and mesh_generator in ("auto", "netgen")
        provider = mesh_provider or default_mesh_provider(scale=mesh_scale)

        simulation_depths, batches = self._prepare_simulation_depths_and_tasks(measurement_depths, batch_size)
        borehole_geometry = np.ascontiguousarray(self.borehole_model[:, :2])
        mud = np.interp(simulation_depths, self.borehole_model[:, 0], self.borehole_model[:, 2])
        if verbose and sweep.rank() == 0:
            print("{} simulation tasks prepared".format(len(batches)))
        # ONE dictionary of solver keywords: the explicit arguments, over-ridden by solver_options (a key given in both places used
        # to raise "multiple values" inside every batch, i.e. a sweep of NaNs); unknown keys fail here, before any batch is drawn,
        # on every rank alike
        base_kw = dict(preconditioner=preconditioner, condense=condense, rtol=rtol, maxsteps=maxsteps, precision=precision)
        base_kw.update(extra)
        opts = solver.make_opts(**base_kw)
        # the solver of the P1 block follows the mesh kind and the number of contexts (vertex_solver_options) unless the caller chose one
        tuned_coarse = base_kw["preconditioner"] == "multigrid" and not any(k in extra for k in ("coarse", "coarse_degree", "coarse_ratio"))
        conforming_default = mesh_provider is None     # the default 3D provider = conforming revolved meshes

        n_tools = len(self.tools)
        results = np.zeros((len(measurement_depths), n_tools))
        t_solve = t_mesh = 0.0
        n_points = 0

        def window(bi):
            if netgen_path:   # the reference's default 2D windowing (remo3d.py:776-779, worker.py:94)
                return geometry.select_netgen_data_range(borehole_geometry, self.formation_model, mud[bi], simulation_depths[bi], domain_radius)
            # Gmsh-path windowing (worker.py:84), the only one for dipping models
            return geometry.select_data_range(borehole_geometry, self.formation_model, self.dip_rad if is3d else 0, mud[bi],
                                              simulation_depths[bi], domain_radius)

        # Which batches this rank takes: its block-cyclic share ("static"), or whatever it draws from the shared counter
        # while it is free ("dynamic", the reference's pull scheduling, remo3d.py:843-860) - sweep.BatchQueue.
        # mesh_workers > 0: the default mesher runs ahead of the solver in that many spawned processes (the
        # reference meshes inside its MPI workers, i.e. in parallel too: worker.py:84-97)
        import collections
        import queue
        import threading
        bq = sweep.BatchQueue(len(batches), schedule)
        share_len = len(list(sweep.my_share(len(batches))))
        pool, pending = None, {}
        if mesh_workers is None:    # default: the reference's cpu_workers mesh (and solve) in parallel; here they mesh, for sweeps long
            mesh_workers = min(int(self.cpu_workers or 0), 8) if (mesh_provider is None and share_len >= 16) else 0   # enough to pay for the start-up
        if mesh_workers > 0 and mesh_provider is None and share_len > 1:
            import sys
            main_mod = sys.modules.get("__main__")
            hidden = {}
            try:
                import multiprocessing
                from concurrent.futures import ProcessPoolExecutor
                # spawned children re-import the caller's __main__ (and would re-run a script without an
                # `if __name__ == "__main__"` guard, GPU contexts and all): hide its path while the workers start -
                # the worker functions live in this module, nothing of __main__ is needed over there
                for attr in ("__file__", "__spec__"):
                    if main_mod is not None and getattr(main_mod, attr, None) is not None:
                        hidden[attr] = getattr(main_mod, attr)
                        setattr(main_mod, attr, None)
                pool = ProcessPoolExecutor(max_workers=int(mesh_workers), mp_context=multiprocessing.get_context("spawn"),
                                           initializer=_mesh_worker_init, initargs=(dict(scale=mesh_scale),))
                # the executor starts a process per submit while none is idle: start them ALL here, while __main__ is hidden
                for fut in [pool.submit(time.sleep, 0.25) for _ in range(int(mesh_workers))]:
                    fut.result()
            except Exception:     # no worker processes here: mesh inline
                pool, pending = None, {}
            finally:
                for attr, v in hidden.items():
                    setattr(main_mod, attr, v)
        ctxs = [self.ctx] + list(getattr(self, "extra_ctx", []))
        n_ctx_total = len(ctxs)
        free_ctx = queue.Queue()
        for c in ctxs:
            free_ctx.put(c)
        acc = dict(mesh=0.0, solve=0.0, points=0, failed_batches=0, not_converged=0, first_error=None, pcg_steps=0, programming_error=None)
        lock = threading.Lock()
        # batches drawn ahead of the solver so that their meshes are in the making: the whole share at once when it is
        # fixed anyway, a few (they are OWNED once drawn) under the pull schedule
        ahead = collections.deque()
        depth = 1 if pool is None else (share_len if schedule == "static" else int(mesh_workers) + len(ctxs))
        draw, draw_lock = iter(bq), threading.Lock()

        def next_batch():
            with draw_lock:
                # under the pull schedule a drawn batch is OWNED: near the end of the sweep a rank must not sit on a queue of
                # batches the other ranks are idle for - draw ahead no further than its fair part of what is left
                cap = depth if schedule == "static" else max(1, min(depth, 1 + bq.remaining_hint() // (2 * max(1, sweep.world_size()))))
                while len(ahead) < cap:
                    try:
                        bi = next(draw)
                    except StopIteration:
                        break
                    if pool is not None:
                        try:
                            fg, bh, _ = window(bi)
                            pending[bi] = pool.submit(_mesh_worker_run, (dim, domain_radius, batches[bi].electrodes, fg, bh, self.dip_rad))
                        except Exception:
                            pass      # reported when the batch's turn comes
                    ahead.append(bi)
                return ahead.popleft() if ahead else None

        def run_batch(bi):
            batch = batches[bi]
            rows = [(r.depth_index, r.tool_index) for s in batch.solves for r in s.records]
            try:
                t0 = time.time()
                fg, bh, sigma = window(bi)
                mesh = None
                if bi in pending:
                    try:
                        mesh = pending.pop(bi).result()
                    except Exception:     # a worker process died: mesh this batch here
                        mesh = None
                if mesh is None:
                    mesh = provider(dim, domain_radius, batch, fg, bh, self.dip_rad)
                sources, evals, readers = tasks.batch_rhs(batch, self.tools)
                t1 = time.time()
                bopts = opts
                if tuned_coarse and getattr(mesh, "dim", 0) == 3:
                    vs = vertex_solver_options(3, mesh.n_nodes, conforming_default, n_ctx_total)
                    if vs:
                        bopts = solver.make_opts(**dict(base_kw, **vs))
                c = free_ctx.get()
                try:
                    outs, st, rc = c.solve_batch(mesh, sigma, sources, evals, bopts)
                finally:
                    free_ctx.put(c)
                t2 = time.time()
                n = 0
                for u, rd in zip(outs, readers):
                    for (di, ti, K, o, m) in rd:
                        results[di, ti] = tasks.apparent_resistivity(u[o:o + m], m, K, dim)
                        n += 1
                with lock:
                    acc["mesh"] += t1 - t0; acc["solve"] += t2 - t1; acc["points"] += n
                    acc["not_converged"] += int(rc == solver.REMO_NOT_CONVERGED); acc["pcg_steps"] += int(st.get("pcg_steps", 0))
            except Exception as ex:
                for di, ti in rows:      # any failure in a batch -> NaN for its records (worker.py:135-138: a bare except)
                    results[di, ti] = np.nan
                with lock:               # ... but not silently: the reference's worker at least shows it on stderr
                    acc["failed_batches"] += 1
                    if acc["first_error"] is None:
                        acc["first_error"] = "batch {}: {}: {}".format(bi, type(ex).__name__, ex)
                    # a programming error (e.g. in a custom mesh_provider) is recorded like any other failure HERE - a rank that
                    # raised now would miss the collectives below and leave the other ranks waiting in the all-reduce - and is
                    # raised once they are through
                    if isinstance(ex, (TypeError, AttributeError, NameError)) and acc["programming_error"] is None:
                        acc["programming_error"] = ex

        def drive():
            while True:
                bi = next_batch()
                if bi is None:
                    return
                run_batch(bi)

        t_busy = time.time()
        try:
            if len(ctxs) > 1 and share_len > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=len(ctxs)) as tp:     # one host thread per context; ctypes calls release the GIL
                    for fut in [tp.submit(drive) for _ in ctxs]:
                        fut.result()
            else:
                drive()
        finally:
            if pool is not None:     # whatever happened above, the mesh processes do not outlive the sweep
                pool.shutdown(wait=False, cancel_futures=True)
        t_busy = time.time() - t_busy
        mine = list(bq.taken)
        t_mesh, t_solve, n_points = acc["mesh"], acc["solve"], acc["points"]
        bq.check_complete()          # collective: every batch was taken exactly once over the ranks
        results = sweep.combine(results)
        self.logs = {name: np.vstack([measurement_depths, results[:, i]]).T for i, name in enumerate(self.tools.keys())}
        self.timing = dict(total_s=time.time() - start, mesh_s=t_mesh, solve_s=t_solve, points=n_points, batches=len(batches),
                           my_batches=len(mine), world_size=sweep.world_size(), schedule=schedule, busy_s=t_busy,
                           busy_s_per_rank=[b[0] for b in sweep.gather_floats([t_busy])], failed_batches=acc["failed_batches"],
                           not_converged=acc["not_converged"], first_error=acc["first_error"], pcg_steps=acc["pcg_steps"])
        if acc["programming_error"] is not None:     # every rank is through the collectives: now it may raise
            raise acc["programming_error"]
        if verbose and acc["failed_batches"]:
            print("rank {}: {} of {} batches failed (NaN in the logs); first: {}".format(sweep.rank(), acc["failed_batches"], len(mine), acc["first_error"]))
        if verbose and acc["not_converged"]:
            print("rank {}: PCG stopped at maxsteps in {} batches".format(sweep.rank(), acc["not_converged"]))
        if verbose and sweep.rank() == 0:
            print("\nProcessed in: ", datetime.timedelta(seconds=self.timing["total_s"]))

    # -- results (remo3d.py:902-1147): Results_<n>.txt per group of logs on one depth grid + Results_plot.png ------------------
    def save_results(self, output_folder=None, measurements_to_save="auto", plot_layout="auto", plot_depth_lim="auto", plot_aspect_ratio="auto",
                     model_rad_lim="auto", model_res_lim="auto", logs_res_lim="auto", logs_at_nan="break", logs_interpolation_factor=1,
                     logs_colours="auto", plot=True):
        """Same keywords as the reference (remo3d.py:902-903).  plot=False (this build's addition) skips the picture; without an
        output folder the reference only shows the figure (a notebook feature): here the figure is returned."""
        plot_kw = dict(plot_layout=plot_layout, plot_depth_lim=plot_depth_lim, plot_aspect_ratio=plot_aspect_ratio, model_rad_lim=model_rad_lim,
                       model_res_lim=model_res_lim, logs_res_lim=logs_res_lim, logs_at_nan=logs_at_nan,
                       logs_interpolation_factor=logs_interpolation_factor, logs_colours=logs_colours)
        if output_folder is None:
            if plot and self.logs and self.formation_model is not None:
                from . import plotting
                return plotting.plot_results(self, None, **plot_kw)
            return None
        sub = os.path.join(output_folder, "Results_{}/".format(datetime.datetime.now().strftime("%Y_%m_%d__%H_%M_%S")))
        os.makedirs(sub, exist_ok=True)
        pending = list(self.logs.keys()) if measurements_to_save == "auto" else list(measurements_to_save)
        n = 1
        written = []
        while pending:
            head = pending[0]
            group = [head] + [k for k in pending[1:] if self.logs[k].shape[0] == self.logs[head].shape[0]
                              and np.all(np.isclose(self.logs[head][:, 0], self.logs[k][:, 0]))]
            pending = [k for k in pending if k not in group]
            table = np.hstack([self.logs[head]] + [self.logs[k][:, 1:2] for k in group[1:]])
            header = "\t".join(["DEPTH"] + group) + "\n" + "\t".join(["M"] + ["OHMM"] * len(group))
            path = sub + "Results_{}.txt".format(n)
            np.savetxt(path, table, fmt="%.4f", delimiter="\t", header=header, comments="")
            written.append(path)
            n += 1
        if plot and self.logs and self.formation_model is not None:     # (no picture of a model that was never set)
            from . import plotting
            import matplotlib.pyplot as plt
            fig = plotting.plot_results(self, sub + "Results_plot.png", **plot_kw)
            plt.close(fig)
            written.append(sub + "Results_plot.png")
        return written
