import cProfile, io, os, pstats, sys, runpy
sys.argv = ["cli_many_contigs.py", "100000", "1"]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
# run the tool once normally (warm), then profile main() again on the same files
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_many_contigs.py")).read()
g = {"__name__": "__main__", "__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "cli_many_contigs.py")}
exec(compile(src, g["__file__"], "exec"), g)
pr = cProfile.Profile(); pr.enable()
g["main"](["predict", g["mpath"], g["fa"], "--output", os.path.join(g["d"], "out.tsv")])
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue())
