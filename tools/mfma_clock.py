"""dl_probe_mfma_sustained looped for a few seconds on random / zero operand bits while rocm-smi samples sclk and socket power."""
import ctypes as C, os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepliif_amd import _lib as L
lib = L.load()
dev = 'cuda'
blocks, iters = 256, 240
n = int(lib.dl_probe_mfma_sustained_elems(blocks))
sink = torch.zeros(4, device=dev)
for tag in ('random', 'zero', 'random'):
    data = (torch.randn(n, device=dev) if tag == 'random' else torch.zeros(n, device=dev)).to(torch.bfloat16)
    samples, stop = [], False

    def sampler():
        while not stop:
            try:
                o = subprocess.run(['rocm-smi', '-c', '-P'], capture_output=True, text=True, timeout=5).stdout
                sclk = re.search(r'sclk clock level.*?\((\d+)Mhz\)', o)
                pw = re.search(r'Power \(W\):\s*([\d.]+)', o)
                samples.append((sclk.group(1) if sclk else '?', pw.group(1) if pw else '?'))
            except Exception as e:
                samples.append(('err', str(e)[:30]))
            time.sleep(0.2)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); launches = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 2.5:
        for _ in range(100):
            lib.dl_probe_mfma_sustained(C.c_void_p(data.data_ptr()), blocks, iters, C.c_void_p(sink.data_ptr()), st)
        launches += 100
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop = True; th.join()
    tf = blocks * 4 * iters * 64 * 32768.0 * launches / (e0.elapsed_time(e1) * 1e-3) / 1e12
    print(tag, 'TF/s', round(tf, 1), 'samples (sclk MHz, W):', samples[1:8])
