"""GPU clock / power under the device-resident loop and under the host loop (sysfs hwmon, sampled at ~50 Hz)."""
import glob, importlib, os, sys, threading, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
spa = importlib.import_module('superpixel-align_amd')
pipeline = importlib.import_module('superpixel-align_amd.pipeline')
drn = importlib.import_module('superpixel-align_amd.drn')
bench = importlib.import_module('bench')
B, H, W = 30, 1024, 2048
args = types.SimpleNamespace(superpixel_method='slic', n_slic_segments=200, n_anchors=10, n_neighbors=4, without_pos=False,
                             y_rel_pos=0.75, x_rel_pos=0.5, y_rel_sigma=0.1, x_rel_sigma=0.1, gpu=0, n_clusters=2,
                             use_feature_maps=[7], pool_mode='mean', mean_sampling='nearest')
model = drn.create_drn('drn_d_22', None, device='cuda', dtype=torch.float32)
pipe = pipeline.LabelPipeline(args, model, overlap=False)
pin = torch.empty((B, 3, H, W), dtype=torch.float32).pin_memory()
bench.make_batch(spa.synth, B, H, W, out=pin.numpy(), integer=True)
dev = pin.cuda()
host = torch.empty((B, H, W, 3), dtype=torch.uint8).pin_memory()
host.copy_(pin.permute(0, 2, 3, 1))
fq = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input'))
pw = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average'))
stop = [False]
samples = []
def sampler():
    while not stop[0]:
        row = []
        for p in fq + pw:
            try:
                row.append(int(open(p).read()))
            except Exception:
                row.append(0)
        samples.append(row)
        time.sleep(0.02)
def measure(label, fn):
    del samples[:]
    stop[0] = False
    th = threading.Thread(target=sampler); th.start()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    stop[0] = True; th.join()
    import numpy as np
    a = np.array(samples[2:], dtype=np.float64)
    k = int(a[:, :len(fq)].mean(0).argmax())
    print('%-16s %.1f ms per batch | busiest card sclk mean %.0f MHz min %.0f max %.0f | power mean %.0f W (%d samples)'
          % (label, dt * 1e3 / 12, a[:, k].mean() / 1e6, a[:, k].min() / 1e6, a[:, k].max() / 1e6,
             a[:, len(fq) + k].mean() / 1e6 if len(pw) > k else -1, len(a)))
hs = pipeline.HostStream(pipe, B, H, W, u8_hwc=True)
def dev_loop():
    for _ in range(12):
        pipe.run(dev, check_status=False)
def host_loop():
    for _ in hs.process(iter([host] * 12)):
        pass
for _ in range(2):
    dev_loop(); host_loop()
for rep in range(2):
    measure('device resident', dev_loop)
    measure('host loop', host_loop)
