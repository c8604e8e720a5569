import cProfile, pstats, sys, io, runpy
sys.argv = ["ingest_bench.py", "--samples", "800", "--clouds", "2000"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("tools/ingest_bench.py", run_name="__main__")
finally:
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print(s.getvalue()[:9000])
