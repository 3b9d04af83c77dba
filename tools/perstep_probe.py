"""Per-step decode times from a cold start and after 0.5 s of idle: shows the ramp of the shader clock (about seven steps,
30 ms) that bench.py keeps out of the timed region with its untimed pre-warm steps."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, rtlsdr_ft8d_amd as ft8
from rtlsdr_ft8d_amd import workload
B=4096
dec=ft8.Decoder(device=0,max_frames=B)
_,tones=workload.message_pool(); sig,_=workload.frame_signals(0,B,20,tones)
iq=torch.empty((B,2,48000),dtype=torch.float32,device='cuda'); dec.synth_frames(sig,B,20,1.0,workload.SEED_BASE,iq)
spots=torch.zeros((B,1400),dtype=torch.uint8,device='cuda'); nres=torch.zeros((B,),dtype=torch.int32,device='cuda'); torch.cuda.synchronize()
ext=torch.cuda.ExternalStream(dec.stream_handle())
ev=[torch.cuda.Event(enable_timing=True) for _ in range(31)]
ev[0].record(ext)
for i in range(30):
    dec.decode_batch_dev(iq,B,spots,nres); ev[i+1].record(ext)
dec.synchronize()
print([round(ev[i].elapsed_time(ev[i+1]),3) for i in range(30)])
time.sleep(0.5)
ev[0].record(ext)
for i in range(10):
    dec.decode_batch_dev(iq,B,spots,nres); ev[i+1].record(ext)
dec.synchronize()
print("after 0.5 s idle:", [round(ev[i].elapsed_time(ev[i+1]),3) for i in range(10)])
