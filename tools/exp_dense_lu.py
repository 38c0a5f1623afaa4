import torch, time
dev = torch.device("cuda")
for n in (1000, 2240, 4032, 6000, 8064):
    try:
        A = torch.randn(n, n, dtype=torch.complex128, device=dev) + n * torch.eye(n, dtype=torch.complex128, device=dev)
        b = torch.randn(n, 1, dtype=torch.complex128, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        LU, piv = torch.linalg.lu_factor(A)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        x = torch.linalg.lu_solve(LU, piv, b)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        for _ in range(20): x = torch.linalg.lu_solve(LU, piv, b)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        inv_t0 = time.perf_counter(); Ai = torch.linalg.inv(A); torch.cuda.synchronize(); inv_t = time.perf_counter() - inv_t0
        y = Ai @ b; torch.cuda.synchronize(); t4 = time.perf_counter()
        for _ in range(20): y = Ai @ b
        torch.cuda.synchronize(); t5 = time.perf_counter()
        print(n, "lu_factor %.1f ms (2nd call below), first solve %.2f ms, solve %.3f ms each, inv %.1f ms, matvec %.3f ms, err %.1e" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)/20, 1e3*inv_t, 1e3*(t5-t4)/20, float((A@x-b).abs().max())))
        t0 = time.perf_counter(); LU, piv = torch.linalg.lu_factor(A); torch.cuda.synchronize(); print("   lu_factor again %.1f ms" % (1e3*(time.perf_counter()-t0)))
    except Exception as e:
        print(n, "FAILED", repr(e)[:200])
