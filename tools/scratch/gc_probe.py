import sys, os, gc, time, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
if mode == "nogc":
    gc.disable()
elif mode == "freeze":
    pass
t = []
gc.callbacks.append(lambda phase, info: t.append((phase, info["generation"], time.perf_counter())))
sys.argv = ["host_time_supernet.py", "3"]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host_time_supernet.py"), run_name="__main__")
# durations of gen-2 collections
starts = {}
long = []
for phase, g, ts in t:
    if phase == "start":
        starts[g] = ts
    else:
        d = ts - starts.get(g, ts)
        if d > 0.005:
            long.append((g, round(d * 1e3, 1)))
print(mode, "collections > 5 ms:", long[-12:], "count", len(long))
