import torch, ctypes, os, sys
here=os.path.dirname(os.path.abspath(__file__))
lib=ctypes.CDLL(os.path.join(here,"libprobe.so"))
maps=[l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l]
print("hip runtimes mapped:", sorted(set(maps)))
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
x=torch.randn(1000,device="cuda"); y=torch.randn(1000,device="cuda"); y0=y.clone()
s=torch.cuda.current_stream().cuda_stream
rc=lib.probe_axpy(ctypes.c_void_p(x.data_ptr()),ctypes.c_void_p(y.data_ptr()),ctypes.c_float(2.0),1000,ctypes.c_void_p(s))
torch.cuda.synchronize(); print("axpy rc",rc,"err",(y-(2*x+y0)).abs().max().item())
K=64
A=torch.randn(32,K,device="cuda"); Bt=torch.randn(32,K,device="cuda"); C=torch.zeros(32,32,device="cuda")
rc=lib.probe_mfma(ctypes.c_void_p(A.data_ptr()),ctypes.c_void_p(Bt.data_ptr()),ctypes.c_void_p(C.data_ptr()),K,ctypes.c_void_p(s))
torch.cuda.synchronize(); ref=(A.double()@Bt.double().T).float(); print("mfma rc",rc,"err",(C-ref).abs().max().item())
# graph capture with foreign launches
g=torch.cuda.CUDAGraph(); y.copy_(y0)
st=torch.cuda.Stream()
with torch.cuda.stream(st):
    with torch.cuda.graph(g, stream=st):
        s2=torch.cuda.current_stream().cuda_stream
        lib.probe_axpy(ctypes.c_void_p(x.data_ptr()),ctypes.c_void_p(y.data_ptr()),ctypes.c_float(2.0),1000,ctypes.c_void_p(s2))
y.copy_(y0); g.replay(); g.replay(); torch.cuda.synchronize()
print("graph err",(y-(4*x+y0)).abs().max().item())
