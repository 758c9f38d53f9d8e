import cProfile, pstats, sys, os, io
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_cli
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
bench_cli.measure(n, reps=1)  # warm: builds, page cache
pr = cProfile.Profile(); pr.enable()
r = bench_cli.measure(n, reps=1)
pr.disable()
print({k: r[k] for k in ('select_main_s', 'map_main_s')})
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
