import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from recpack_amd.devmon import DeviceMonitor, _hwmon_dirs, pci_address_of
print(pci_address_of(0)); print(_hwmon_dirs()[:20])
p = torch.cuda.get_device_properties(0); print(p)
m = DeviceMonitor(0).start()
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
t0=time.time()
while time.time()-t0 < 3: b = a @ a
torch.cuda.synchronize(); m.stop(); print(m.summary())
