"""time of diee_train_wgrad3x3 (k_wgrad3x3 + k_wgrad_fold) at batch 256 with HIP events; DIEE_LIB selects the build"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diee_amd
L = diee_amd.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B * 24, 256, device="cuda").bfloat16(); dy = torch.randn(B * 24, 256, device="cuda").bfloat16()
dw = torch.empty(256, 256, 3, 3, device="cuda"); scr = torch.empty(int(L.diee_train_wgrad_scratch_floats()), device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(n):
    for _ in range(n):
        L.diee_train_wgrad3x3(C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), B, C.c_void_p(scr.data_ptr()), st)
run(20); torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record(); run(200); b.record(); torch.cuda.synchronize()
print(f"{os.environ.get('DIEE_LIB', 'default')}: wgrad + fold {a.elapsed_time(b) / 200 * 1e3:.1f} us per call (batch {B})")
