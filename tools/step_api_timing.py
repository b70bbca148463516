import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import ssme_amd
y = np.loadtxt(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests', 'golden', 'spy_returns.csv'))[:400]
for (model, n, r, th) in ((0, 500, 1, [1.0, .95, .25]), (0, 65536, 1, [1.0, .95, .25]), (1, 16384, 512, [0.9, 0.0, 1.0, -0.1]), (0, 1 << 20, 1, [1.0, .95, .25])):
    b = ssme_amd.ParticleFilterBank(model, n, r, 1)
    b.set_params(th)
    for t in range(20): b.step(y[t], 0.0 if model == 1 else None)
    t0 = time.perf_counter()
    for t in range(20, 220): b.step(y[t], 0.0 if model == 1 else None)
    dt = (time.perf_counter() - t0) / 200
    print(f"step API model {model} N={n} R={r}: {dt*1e6:.1f} us per filter() call")
    b.close()
