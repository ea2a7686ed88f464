#!/usr/bin/env python3
"""Round 6 probe (GPU box): what the amdgpu hwmon / sysfs nodes of HIP device 0 offer to an unprivileged process, and how fast they read."""
import glob
import os
import time

import torch

p = torch.cuda.get_device_properties(0)
print("torch props:", {k: getattr(p, k) for k in ("name", "pci_bus_id", "pci_device_id", "pci_domain_id", "multi_processor_count") if hasattr(p, k)})
bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
print("bdf", bdf, os.path.isdir("/sys/bus/pci/devices/" + bdf))
for d in sorted(glob.glob("/sys/bus/pci/devices/*/hwmon/hwmon*")):
    if "amdgpu" not in open(os.path.join(d, "name")).read():
        continue
    vals = {}
    for f in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
        try:
            vals[f] = open(os.path.join(d, f)).read().strip()
        except Exception as e:
            vals[f] = "ERR " + type(e).__name__
    print(d, vals)
d = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % bdf)
if d:
    d = d[0]
    torch.cuda.init()
    x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    t0 = time.time()
    n = 0
    while time.time() - t0 < 1.0:
        y = x @ x
        for f in ("power1_average", "power1_input", "freq1_input"):
            try:
                v = open(os.path.join(d, f)).read().strip()
            except Exception:
                v = None
        n += 1
    torch.cuda.synchronize()
    print("reads per second (3 files each, beside matmuls):", n, "last:", {f: (open(os.path.join(d, f)).read().strip() if os.path.exists(os.path.join(d, f)) else None) for f in ("power1_average", "power1_input", "freq1_input")})
    for f in ("pp_dpm_sclk", "gpu_busy_percent", "current_link_speed"):
        try:
            print(f, open("/sys/bus/pci/devices/%s/%s" % (bdf, f)).read().strip().replace("\n", " | "))
        except Exception as e:
            print(f, "ERR", type(e).__name__)
