import torch, time
dev=torch.device("cuda",0)
for mb in (1,8,32):
    n=mb<<20
    src=torch.empty(n,dtype=torch.uint8,device=dev)
    dst=torch.empty(n,dtype=torch.uint8).pin_memory()
    s=torch.cuda.Stream()
    for rep in range(3):
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            t0=time.perf_counter(); dst.copy_(src,non_blocking=True); t1=time.perf_counter()
            s.synchronize(); t2=time.perf_counter()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            t3=time.perf_counter(); src.copy_(dst,non_blocking=True); t4=time.perf_counter()
            s.synchronize(); t5=time.perf_counter()
    print(mb,"MB  D2H call %.1f us, until done %.1f us | H2D call %.1f us, until done %.1f us"%((t1-t0)*1e6,(t2-t0)*1e6,(t4-t3)*1e6,(t5-t3)*1e6))
