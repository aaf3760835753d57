import ctypes, json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["ARIA_GEMM_FORCE"]="3"
dev="cuda"; bf16=torch.bfloat16
root=os.environ.get("GRAFT_REPO_ROOT",".")
libs={"full": os.path.join(root,"aria_amd","libaria_hip.so"), "no C store": os.path.join(root,"build","abl","libgemm_abl64.so")}
res={}
M,N=16384,8192
for K in (640,1152,2560):
    x=torch.randn(M,K,device=dev).to(bf16); w=(torch.randn(N,K,device=dev)*0.02).to(bf16); out=torch.empty(M,N,dtype=bf16,device=dev)
    for pers in ("0","1"):
        os.environ["ARIA_GEMM_PERSIST"]=pers
        for name,path in libs.items():
            lib=ctypes.CDLL(path); fn=lib.aria_gemm_bf16
            fn.argtypes=[ctypes.c_void_p]*4+[ctypes.c_int64]*3+[ctypes.c_int]*2+[ctypes.c_int64]*3+[ctypes.c_int]*2+[ctypes.c_void_p]
            st=torch.cuda.current_stream().cuda_stream
            call=lambda: fn(x.data_ptr(),w.data_ptr(),out.data_ptr(),None,M,N,K,0,0,K,K,N,0,0,st)
            for _ in range(3): call()
            torch.cuda.synchronize()
            s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): call()
            e.record(); torch.cuda.synchronize()
            res[f"K{K} persist={pers} {name}"]=round(s.elapsed_time(e)/20*1e3/8,2)
print(json.dumps(res))
