"""How fast are the library f32 GEMMs (torch.matmul -> rocBLAS / hipBLASLt) on the trainer's shapes?  A yardstick for
k_gemm_f32 (csrc/az_train.hip); not used by the product."""
import time, torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda")
shapes = {"conv2 fwd  [2688x4608]x[4608x512]": (2688, 4608, 512, "nn"),
          "conv2 dgrad [2688x512]x[512x4608]": (2688, 512, 4608, "nt"),
          "conv2 wgrad [4608x2688]x[2688x512]": (4608, 2688, 512, "tn"),
          "conv3 fwd  [1280x4608]x[4608x512]": (1280, 4608, 512, "nn"),
          "conv4 fwd  [384x4608]x[4608x512]": (384, 4608, 512, "nn"),
          "fc1 fwd    [64x3072]x[3072x1024]": (64, 3072, 1024, "nn"),
          "fc1 wgrad  [3072x64]x[64x1024]": (3072, 64, 1024, "tn")}
for name, (M, K, N, mode) in shapes.items():
    if mode == "nn":
        A = torch.randn(M, K, device=dev); B = torch.randn(K, N, device=dev); f = lambda: A @ B
    elif mode == "nt":
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); f = lambda: A @ B.t()
    else:
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev); f = lambda: A.t() @ B
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(50): f()
    torch.cuda.synchronize(); dt = (time.time() - t) / 50
    print(f"{name:40s} {dt*1e6:8.1f} us  {2*M*N*K/dt/1e12:6.1f} TFLOP/s")
