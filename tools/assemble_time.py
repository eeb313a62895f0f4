import os as _os
# the MNV_* knobs this tool reads exist in the test-hook build of the library only (csrc/mnv_knobs.h)
_os.environ.setdefault("MNV_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mega-nerf-viewer_amd", "testhooks", "libmnv.so"))
import os, sys, time
sys.path[:0]=[os.getcwd(), os.path.join(os.getcwd(),"tests")]
import torch, mega_nerf_viewer_amd as mnv
from mega_nerf_viewer_amd.multigpu import TilePartition
W,H,NF,world=1920,1080,64,8
part=TilePartition(W,H,world,64,24,6)
g=torch.randint(0,255,(world,NF,part.j_max,24,64,4),dtype=torch.uint8,device="cuda")
out=torch.empty((NF,H,W,4),dtype=torch.uint8,device="cuda")
want=part.unpermute(g)   # torch index path
n_cus=torch.cuda.get_device_properties(0).multi_processor_count
for reserve in (0, n_cus-32):
    h,en=mnv.stream_create_reserved(reserve); st=torch.cuda.ExternalStream(h)
    with torch.cuda.stream(st):
        for _ in range(3): part.unpermute(g,out=out)
        st.synchronize(); t0=time.perf_counter()
        for _ in range(20): part.unpermute(g,out=out)
        st.synchronize(); ms=(time.perf_counter()-t0)/20*1e3
    print("narrow" if os.environ.get("MNV_ASSEMBLE_NARROW") else "wide", "enabled CUs", en, "ms %.3f"%ms, "GB/s %.0f"%(2*out.numel()/ms/1e6), "equal", bool(torch.equal(out,want)))
