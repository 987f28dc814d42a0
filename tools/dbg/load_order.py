import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
if mode == "lib_only":
    from gs_localization_amd import _lib
    print("lib only:", _lib.load().gsr_device_ok())
elif mode == "lib_then_torch":
    from gs_localization_amd import _lib
    lib = _lib.load()
    import torch
    print("torch avail", torch.cuda.is_available(), "ok", lib.gsr_device_ok(), lib.gsr_last_error())
elif mode == "torch_then_lib":
    import torch
    print("torch avail", torch.cuda.is_available())
    from gs_localization_amd import _lib
    print("ok", _lib.load().gsr_device_ok())
os.system("grep -E 'libamdhip64|libhsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
