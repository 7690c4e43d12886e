"""(box) What the driver's metrics say while the headline kernel runs: raw gpu_metrics blob (header + hexdump), amd-smi / rocm-smi JSON under load.
Run: python tools/probe_metrics.py > gpurun_out/probe_metrics.txt"""
import glob, json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from phonic_amd import workloads
from phonic_amd.graph import Graph

g = Graph(48000, 2, 1024, 0)
g.set_max_blocks_per_launch(32)
workloads.build_headline(g, 1024)
st = torch.cuda.Stream()
bus = torch.zeros(32 * 2048, device="cuda:0")
stop = False
def load():
    pos = 0
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(8):
                g.write_device(bus.data_ptr(), bus.numel(), pos, st.cuda_stream)
                pos += 32 * 1024
            torch.cuda.synchronize()
t = threading.Thread(target=load); t.start()
time.sleep(2.0)
pr = torch.cuda.get_device_properties(0)
want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
card = [d for d in glob.glob("/sys/class/drm/card*/device") if os.path.basename(os.path.realpath(d)).startswith(want)][0]
print("card", card)
for k in range(3):
    blob = open(card + "/gpu_metrics", "rb").read()
    print("gpu_metrics len", len(blob), "header size", int.from_bytes(blob[0:2], "little"), "format", blob[2], "content", blob[3])
    print(blob.hex())
    time.sleep(0.5)
for cmd in (["amd-smi", "metric", "--json"], ["amd-smi", "metric", "--clock", "--power", "--usage", "--json"], ["rocm-smi", "--showclocks", "--showpower", "--showuse", "--showmemuse", "--json"], ["amd-smi", "version"]):
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        print("==", " ".join(cmd), "rc", out.returncode)
        print(out.stdout[:20000])
        print(out.stderr[:1000])
    except Exception as e:
        print("==", cmd, "failed", e)
stop = True
t.join()
