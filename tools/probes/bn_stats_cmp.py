import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
for k in a.files:
    if "num_batches" in k: continue
    d = np.abs(a[k] - b[k]) / np.maximum(1e-12, np.abs(b[k]))
    print("%-45s max per-channel rel diff %.3e (at %d: %.8g vs %.8g)" % (k, d.max(), d.argmax(), a[k][d.argmax()], b[k][d.argmax()]))
