import sys, ctypes as C, os
order = sys.argv[1]
sys.path.insert(0, os.getcwd())
if order == "torch_first":
    import torch
    print("torch avail", torch.cuda.is_available())
import our_first_climate_model_amd as pkg
try:
    e = pkg.Ensemble(4); print(order, "create ok"); e.close()
except Exception as ex:
    print(order, "FAILED", ex)
if order == "lib_first":
    import torch
    print("torch avail after", torch.cuda.is_available())
    try:
        e = pkg.Ensemble(4); print(order, "create-after-torch ok"); e.close()
    except Exception as ex:
        print(order, "FAILED after torch", ex)
os.system(f"grep -E 'amdhip|hsa-runtime' /proc/{os.getpid()}/maps | awk '{{print $6}}' | sort -u")
