import torch, time
import torch.nn.functional as F
B,H,N,D=256,3,197,64
q,k,v=(torch.randn(B,H,N,D,device="cuda",dtype=torch.bfloat16,requires_grad=True) for _ in range(3))
def manual():
    a=((q@k.transpose(-2,-1))*D**-0.5).softmax(-1)
    return a@v
def sdpa():
    return F.scaled_dot_product_attention(q,k,v)
for name,fn in (("manual",manual),("sdpa",sdpa)):
    try:
        for _ in range(3):
            o=fn(); o.sum().backward()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(20):
            o=fn(); o.sum().backward()
        torch.cuda.synchronize(); print(name,(time.perf_counter()-t0)/20*1e3,"ms fwd+bwd")
    except Exception as e:
        print(name,"failed",repr(e)[:300])
print((manual()-sdpa()).abs().max().item())
