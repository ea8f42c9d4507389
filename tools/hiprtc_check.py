"""Compiles a device-source .inc with hipRTC for gfx950 on a box WITHOUT a GPU (what infera_load_model does at run time),
so hipRTC-only compile errors (no <cstdint>, no <cmath> macros) show up before a GPU call.
usage: python tools/hiprtc_check.py infera_amd/csrc/hip/chain_device.inc '<name expression>' [...]"""
import ctypes, sys
rtc = ctypes.CDLL("/opt/rocm/lib/libhiprtc.so")
src = open(sys.argv[1], "rb").read()
prog = ctypes.c_void_p()
assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src, b"check.hip", 0, None, None) == 0
for e in sys.argv[2:]:
    rtc.hiprtcAddNameExpression(prog, e.encode())
opts = (ctypes.c_char_p * 3)(b"--offload-arch=gfx950", b"-O3", b"-std=c++17")
rc = rtc.hiprtcCompileProgram(prog, 3, opts)
n = ctypes.c_size_t()
rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
log = ctypes.create_string_buffer(n.value or 1)
rtc.hiprtcGetProgramLog(prog, log)
print("rc", rc, log.value.decode()[:3000])
sys.exit(rc != 0)
