import sys, time, faulthandler
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
faulthandler.dump_traceback_later(75, exit=True)
import bench, os
os.environ["REMO_BENCH_TRACE_MESH"]="1"
t0=time.time()
w = bench._build_some((100, 1.2, 'lattice', [int(sys.argv[1]) if len(sys.argv) > 1 else 1]))
print(w[0]['mesh'].n_elems, time.time()-t0)
