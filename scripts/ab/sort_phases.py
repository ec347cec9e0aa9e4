import os,sys,time,numpy as np
sys.path.insert(0,".")
import __graft_entry__ as ge, torch
pkg=ge.load_package(); bbg=pkg.Bbg(0); bbg.set_stream(torch.cuda.current_stream().cuda_stream)
for s in (1,):
    bbg.set_option("msm_sort",s); bbg.set_option("msm_async_reduce",0)
    for lg in (20,):
        n=1<<lg; srs=bbg.srs_synth_hashed(0xBB254,n)
        sc=torch.from_numpy(pkg.synthetic_scalars(7,n).view(np.int64).reshape(-1)).cuda()
        out=torch.zeros(12,dtype=torch.int64,device="cuda")
        for _ in range(3): bbg.msm_device(srs,sc.data_ptr(),n,out.data_ptr())
        bbg.sync(); bbg.profile_enable(True)
        for _ in range(10): bbg.msm_device(srs,sc.data_ptr(),n,out.data_ptr())
        bbg.sync()
        print("sort",s,lg,{k: round(bbg.profile_get(k)[0]/10,4) for k in ("msm_recode","msm_sort","msm_accumulate","msm_reduce")},flush=True)
        bbg.profile_enable(False); srs.free()
