import sys, time, ctypes
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from libde265_amd import capi, synth, worklist
lib=capi.Library(); ctx=capi.Context(lib,0)
cfg=dict(synth.CONFIGS["c5_8k10_8tiles"]); pic=synth.picture(**cfg); pp=pic.pp[0]
refs=[]
for i in range(2):
    f=ctx.frame_create_for(pp); ctx.frame_upload(f, synth.ref_planes(cfg["seed"]+17*i,int(pp["width"]),int(pp["height"]),1,10)); refs.append(f)
pic.dst_frame=ctx.frame_create_for(pp); pic.ref_frames=[refs[i] if i<2 else -1 for i in range(worklist.MAX_REF_FRAMES)]
ctx.set_pipeline_depth(3)
st={}
for _ in range(6): ctx.submit_in_place(pic,state=st)
ctx.wait()
L=lib.lib
tb=th=ts=0; N=200
t0=time.perf_counter()
for _ in range(N):
    a=time.perf_counter(); L.m355_arena_begin(ctx.h, st["a_caps"], st["a_dst"]); b=time.perf_counter()
    st["lib"].m355_synth_fill_arena_header(st["a_src"], st["a_dst"]); st["dst"].dst_frame=st["src"].dst_frame; ctypes.memmove(st["dst"].ref_frames, st["src"].ref_frames, 128); c=time.perf_counter()
    L.m355_submit_picture(ctx.h, st["a_dst"]); d=time.perf_counter()
    tb+=b-a; th+=c-b; ts+=d-c
ctx.wait(); tot=time.perf_counter()-t0
print("per step ms: total %.3f arena_begin %.3f header %.3f submit %.3f" % (1e3*tot/N, 1e3*tb/N, 1e3*th/N, 1e3*ts/N))
