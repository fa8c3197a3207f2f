"""k-NN kernel lab: python tools/knn_lab.py [--B 32 --N 1024 --k 20] NAME=path/to/libknn.so ...
Times svnet_knn_f32 of every given library (variant builds of knn.hip alone: tools/knn_lab_build.sh) for the feature widths of the four
graphs of the SV-DGCNN callers and compares the neighbour lists with the in-tree library's, bit for bit.  Diagnostic."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
args = sys.argv[1:]
B, N, k = 32, 1024, 20
while args and args[0].startswith("--"):
    v = int(args[1])
    if args[0] == "--B": B = v
    elif args[0] == "--N": N = v
    elif args[0] == "--k": k = v
    args = args[2:]
libs = [("default", os.path.join(ROOT, "svnet_amd", "libsvnet_hip.so"))] + [tuple(a.split("=")) for a in args]
c_p, c_i64, c_int, c_sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
torch.manual_seed(0)
dev = torch.device("cuda")
ref = {}
for name, path in libs:
    L = ctypes.CDLL(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    L.svnet_knn_workspace_bytes.restype = c_sz
    L.svnet_knn_workspace_bytes.argtypes = [c_i64, c_i64, c_i64]
    L.svnet_knn_f32.restype = c_int
    L.svnet_knn_f32.argtypes = [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_int, c_p, c_p, c_sz, c_p]
    L.svnet_last_error.restype = ctypes.c_char_p
    res = {}
    for C in (3, 62, 127):
        g = torch.Generator(device="cpu").manual_seed(C)
        x = torch.randn(B, N, C, generator=g).to(dev)                  # rows [B, N, C]: the layout of the feature-space graphs (xx_mode 1)
        if C == 3:
            x = x.transpose(1, 2).contiguous().transpose(1, 2)          # the coordinate graph is channel-first (xx_mode 0)
        sb, sn, sc = x.stride()
        mode = 0 if C == 3 else 1
        ws = torch.empty(L.svnet_knn_workspace_bytes(B, N, C), dtype=torch.uint8, device=dev)
        idx = torch.empty(B, N, k, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        def call():
            rc = L.svnet_knn_f32(x.data_ptr(), B, N, C, sb, sn, sc, mode, k, idx.data_ptr(), ws.data_ptr(), ws.numel(), st)
            assert rc == 0, L.svnet_last_error()
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        same = None
        if name == "default":
            ref[C] = idx.clone()
        else:
            same = int((idx != ref[C]).sum().item())
        res["C=%d" % C] = {"us": round(ts[len(ts) // 2], 1), "min_us": round(ts[0], 1), "differs": same}
    print(json.dumps({"lib": name, "B": B, "N": N, "k": k, **res}), flush=True)
