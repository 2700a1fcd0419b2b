import glob, sqlite3, sys
f = glob.glob(sys.argv[1] + "/**/*results.db", recursive=True)
t = sqlite3.connect(f[0])
for r in t.execute("select name,total_calls,average from top_kernels"):
    if "sgcn" in r[0]: print("%-100s %5d %8.2f us" % (r[0][:100], r[1], r[2]))
