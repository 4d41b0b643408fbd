import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for t in range(a["q"].shape[0]):
    d = np.abs(a["q"][t] - b["q"][t]).max(axis=1)
    bad = np.nonzero(~(d == 0))[0]
    if len(bad):
        waves = np.unique(bad // 64)
        print("step", t, "differing envs", len(bad), "in", len(waves), "waves; first", bad[:12], "max diff %.3e" % np.nanmax(d[bad]), "lanes of first wave", (bad[bad // 64 == waves[0]] % 64)[:20])
        if t > 3: break
else:
    print("bitwise identical over", a["q"].shape[0], "steps")
