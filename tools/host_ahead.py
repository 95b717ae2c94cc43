#!/usr/bin/env python3
"""How far the host runs ahead of the device in the bench's forward step: host time per step() call (enqueue only, no sync) against
the device time per step.  A host share close to 1 means the device waits for launches somewhere in the step."""
import sys, time, torch
sys.path.insert(0, ".")
from neural_marionette_amd import NeuralMarionette, HotPathOptions, synth

def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    opts = HotPathOptions(grid_size=64)
    net = NeuralMarionette(opts); net.load_state_dict(synth.make_state_dict(opts, seed=42, variant="peaky")); net = net.to(dev).eval(); net.anneal(1)
    vox = synth.figure_clip(4, 16, 64, seed=1).to(dev)
    eps = synth.make_eps((16, 10, 4, opts.nlatent_kypt), seed=100).to(dev)
    acts = {"detector": True, "learner": True}
    with torch.no_grad():
        for _ in range(5): net(vox, acts, eps=eps)
        torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for _ in range(steps):
            a = time.perf_counter(); net(vox, acts, eps=eps); host.append(time.perf_counter() - a)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    host.sort()
    print("host per step: median %.2f ms (min %.2f, max %.2f); enqueue of %d steps %.1f ms, device done after %.1f ms (%.2f ms per step, %.0f voxel-frames/s)"
          % (host[len(host) // 2] * 1e3, host[0] * 1e3, host[-1] * 1e3, steps, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e3 / steps, 64 * steps / (t2 - t0)))

if __name__ == "__main__":
    main()
