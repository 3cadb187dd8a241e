import sys, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import pathtracer_amd as pt
from harness import Oracle
name, flags = sys.argv[1], int(sys.argv[2])
sc = pt.Scene.load(f"tests/golden/{name}_scene.npz" if name!="cornell" else "tests/golden/cornell.ptxscene")
z = np.load(f"tests/golden/{name}_paths_f{flags}.npz")
ap = float(z["aperture"]) if "aperture" in z else 0.0
prm = pt.make_params(int(z["sx"]), int(z["depth"]), flags, ap, seed=int(z["seed"]))
lib = C.CDLL("scratch/libcpudev.so"); lib.cpudev_load.restype=C.c_void_p; lib.cpudev_load.argtypes=[C.c_void_p]; lib.cpudev_trace.argtypes=[C.c_void_p]*5+[C.c_int,C.c_void_p,C.c_int]
d = sc.desc(); h = C.c_void_p(lib.cpudev_load(C.byref(d)))
pi,pj,pk = [np.ascontiguousarray(z[k],dtype=np.int32) for k in ("pi","pj","pk")]
out = np.zeros((len(pi),3))
lib.cpudev_trace(h, C.byref(prm), pi.ctypes.data, pj.ctypes.data, pk.ctypes.data, len(pi), out.ctypes.data, 0)
err = np.abs(out - z["col"]).max(axis=1)
bad = np.where(err > 1e-9 + 1e-7*np.abs(z["col"]).max(axis=1))[0]
print(name, flags, "bad", len(bad), "of", len(pi))
for q in bad[:int(sys.argv[3]) if len(sys.argv)>3 else 10]:
    print(q, pi[q],pj[q],pk[q], out[q], z["col"][q])
if len(sys.argv)>4:
    q = int(sys.argv[4])
    o1 = np.zeros((1,3))
    lib.cpudev_trace(h, C.byref(prm), pi[q:q+1].ctypes.data, pj[q:q+1].ctypes.data, pk[q:q+1].ctypes.data, 1, o1.ctypes.data, 1)
    orc = Oracle(sc); orc.lib.orc_set_debug(1)
    print(orc.trace_pixels(prm, pi[q:q+1], pj[q:q+1], pk[q:q+1]), o1)
