import sys, time, torch
sys.path.insert(0, "/root/repo")
from kvzip_amd import score
dummies = [torch.cuda.Stream() for _ in range(int(sys.argv[1]))]
t0 = time.perf_counter()
st = score._side_streams("cuda:0", 2)
print("picked", st, "in", round((time.perf_counter() - t0) * 1e3, 1), "ms; overlap check:", score._overlaps(st[0], st[1]))
