import sys, os, ctypes
sys.path.insert(0, os.getcwd())
mode = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l})
if mode == 'torch_first':
    import torch
    print('avail', torch.cuda.is_available()); print(maps())
    x = torch.zeros(4, device='cuda'); print('torch alloc ok'); print(maps())
    from ceedpetscsolid_amd import ceed as cd
    L = cd.CeedLib(cd.PRODUCT_LIB); print(maps())
    c = cd.Ceed(L, '/gpu/hip/mi355x'); print('ceed ok', c.resource)
elif mode == 'torch_import_only':
    import torch
    print(maps())
    from ceedpetscsolid_amd import ceed as cd
    L = cd.CeedLib(cd.PRODUCT_LIB); print(maps())
    c = cd.Ceed(L, '/gpu/hip/mi355x'); print('ceed ok', c.resource)
    x = torch.zeros(4, device='cuda'); print('torch alloc ok')
else:
    from ceedpetscsolid_amd import ceed as cd
    L = cd.CeedLib(cd.PRODUCT_LIB); print(maps())
    c = cd.Ceed(L, '/gpu/hip/mi355x'); print('ceed ok', c.resource)
    import torch
    x = torch.zeros(4, device='cuda'); print('torch alloc ok'); print(maps())
