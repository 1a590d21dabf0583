import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/scripts") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "instance-segmentation-road-project_amd")]
import numpy as np, torch
from masklab_hip import _lib, ops, packing


def main():
    SH = [("s1 conv3 128->256 +res", (8,256,256),128,256,True), ("s2 conv3 256->512 +res",(8,128,128),256,512,True),
          ("s3 conv3 512->1024 +res",(8,64,64),512,1024,True), ("s4 conv3 1024->2048 +res",(8,32,32),1024,2048,True),
          ("s2 conv1 512->256",(8,128,128),512,256,False), ("s3 conv1 1024->512",(8,64,64),1024,512,False)]
    rng=np.random.default_rng(0)
    ops.set_conv_math("f32x3")
    for label,(B,H,W),cin,cout,res in SH:
        x=torch.from_numpy(rng.normal(size=(B,H,W,cin)).astype(np.float32)).cuda()
        w=(rng.normal(size=(1,1,cin,cout))/np.sqrt(cin)).astype(np.float32); b=rng.normal(size=(cout,)).astype(np.float32)
        r=torch.from_numpy(rng.normal(size=(B,H,W,cout)).astype(np.float32)).cuda() if res else None
        line=f"{label:28s}"
        for t in (1,2,3):
            dc=ops.DeviceConv(packing.pack_dense(w,b,tile=t),"cuda")
            out=ops.conv2d(x,dc,padding="same",act=_lib.ACT_RELU,residual=r)
            best=[]
            for _ in range(3):
                s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10): ops.conv2d(x,dc,padding="same",act=_lib.ACT_RELU,residual=r,out=out)
                e.record(); torch.cuda.synchronize(); best.append(s.elapsed_time(e)/10)
            line+=f"  tile{t}: {min(best)*1e3:7.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
