import importlib, os, sys, time
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
sys.path.insert(0, os.getcwd())
import torch, torch.nn.functional as F
engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
torch.backends.cudnn.benchmark = True
B,Cin,Cout,H,W,dil=30,512,512,128,256,4
x = torch.randn((B, Cin, H, W), device='cuda').to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
w = (torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5).to(torch.bfloat16)
w_cl = w.contiguous(memory_format=torch.channels_last)
bias = torch.randn((Cout,), device='cuda')
wt = w.permute(0, 2, 3, 1).reshape(Cout, 9, Cin).contiguous()
fl = 2.0 * B * H * W * Cout * 9 * Cin
def t(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
y = eng.conv3x3_bf16(x, wt, bias, None, True, dil)
ref = torch.relu(F.conv2d(x[:1].float(), w.float(), bias, 1, dil, dil))
print('rel err', ((y[:1].float()-ref).abs().max()/ref.abs().max()).item())
a = t(lambda: eng.conv3x3_bf16(x, wt, bias, None, True, dil), 50)
b = t(lambda: F.conv2d(x, w_cl, None, 1, dil, dil), 50)
print('own %.3f ms %.0f TF | MIOpen %.3f ms %.0f TF' % (a, fl/a/1e9, b, fl/b/1e9))
