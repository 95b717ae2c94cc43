"""Per-step wall time of the detector-mode training step (64^3, T = 16, B = 4) for a conv mode: how many steps the transient lasts.
usage: time_train_steps.py [mode] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth
from neural_marionette_amd.train import DetectorTrainer
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
o = HotPathOptions(grid_size=64)
net = NeuralMarionette(o)
net.load_state_dict(synth.make_state_dict(o, seed=42, variant="peaky"))
net = net.cuda().train(); net.anneal(1); net.set_conv_mode(mode)
vox = synth.figure_clip(4, 16, 64, seed=77).cuda()
tr = DetectorTrainer(net, lr=4e-4)
ts = []
for i in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(vox, sync=False)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(mode, " ".join("%.1f" % t for t in ts))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(8): tr.step(vox, sync=False)
torch.cuda.synchronize(); print("8 steps back to back: %.2f ms/step" % ((time.perf_counter() - t0) * 1e3 / 8))
