import sys, ctypes
sys.path.insert(0,'.')
order = sys.argv[1]
if order == 'lib_first':
    from madaiemulator_amd import abi
    L = abi.load(); print("devcount(lib)", L.gpemu_device_count())
    import torch; print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
else:
    import torch; print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    from madaiemulator_amd import abi
    L = abi.load(); print("devcount(lib)", L.gpemu_device_count())
    import numpy as np
    from madaiemulator_amd import synth
    ctx = abi.Context(0); X,y = synth.design(300,4,1); ctx.set_model(1,1,X,y); print(ctx.loglik(synth.default_thetas(1,4))['value'])
    torch.cuda.synchronize(); print("sync ok")
