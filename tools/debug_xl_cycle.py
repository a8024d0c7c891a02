import sys
sys.path.insert(0, "/root/repo")
import bench
from remo3d_amd import solver
w = bench.build_workload(0, 1, 20, bench.SIZES["XL"], max_batches=1)["work"][0]
with solver.Context(0) as ctx:
    b = ctx.batch(w["mesh"], w["sigma"], w["sources"], w["evals"])
    rc = b.run(solver.make_opts(rtol=1e-8, coarse="amg"), raise_on_error=False)
    print("XL forced cycle: rc", rc, "|", ctx.last_error(), "| nv", w["mesh"].n_nodes, "steps", b.stats["pcg_steps"], "solve ms", b.stats["ms_solve"], "total", b.stats["ms_total"])
    rc = b.run(solver.make_opts(rtol=1e-8), raise_on_error=False)
    print("XL polynomial: rc", rc, "steps", b.stats["pcg_steps"], "solve ms", b.stats["ms_solve"], "total", b.stats["ms_total"])
