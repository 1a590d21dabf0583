import ctypes as C, numpy as np
lib = C.CDLL("scripts/_ab/lds_probe.so")
out = (C.c_float * 512)()
print("rc", lib.run_probe(out))
a = np.array(out[:]).reshape(2, 64, 4)
print("first load (reversed lanes): lane0..3:", a[0, :4].tolist(), " lane63:", a[0, 63].tolist())
print("second load: lane4..6:", a[1, 4:7].tolist())
