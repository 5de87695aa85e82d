"""profiles/tools/gpu_gaps.py on a hand-made pair of rocprofv3 CSV traces: the idle intervals between kernels of the evolve (first to last marching
sweep) and the blocking HIP call the host returned from before the kernel that ends each of them."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_gaps_attributes_idle_intervals(tmp_path):
    d = tmp_path / "run"
    d.mkdir()
    us = 1000
    kernels = [  # (start, end, name, correlation id)
        (0 * us, 50 * us, "setup_kernel(int)", 1),                     # before the first marching sweep: not part of the evolve
        (1000 * us, 1100 * us, "void k_sweep_march<1, 3>(SweepArgs)", 2),
        (1105 * us, 1200 * us, "void k_sweep_march<2, 3>(SweepArgs)", 3),  # 5 us after the one before: below the threshold
        (1260 * us, 1270 * us, "void k_fluxreg<1>(FrItem const*)", 4),     # 60 us idle, the host came back from hipMemcpy
        (1300 * us, 1310 * us, "k_interp(InterpItem const*)", 5),          # 30 us idle, no blocking call
        (1320 * us, 1400 * us, "void k_sweep_march<2, 3>(SweepArgs)", 6),
        (5000 * us, 5100 * us, "teardown(int)", 7),                        # after the last marching sweep
    ]
    with open(d / "1_kernel_trace.csv", "w") as f:
        f.write("Kind,Agent_Id,Queue_Id,Kernel_Id,Kernel_Name,Correlation_Id,Start_Timestamp,End_Timestamp\n")
        for s, e, n, c in kernels:
            f.write(f'KERNEL_DISPATCH,1,1,1,"{n}",{c},{s},{e}\n')
    api = [  # (start, end, function, correlation id)
        (990 * us, 995 * us, "hipLaunchKernel", 2), (996 * us, 999 * us, "hipLaunchKernel", 3),
        (1000 * us, 1250 * us, "hipMemcpy", 90), (1252 * us, 1255 * us, "hipLaunchKernel", 4),
        (1290 * us, 1295 * us, "hipLaunchKernel", 5), (1296 * us, 1299 * us, "hipLaunchKernel", 6),
    ]
    with open(d / "1_hip_api_trace.csv", "w") as f:
        f.write("Domain,Function,Process_Id,Thread_Id,Correlation_Id,Start_Timestamp,End_Timestamp\n")
        for s, e, n, c in api:
            f.write(f"HIP_RUNTIME_API,{n},1,1,{c},{s},{e}\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "gpu_gaps.py"), str(d), "15"], capture_output=True, text=True, check=True).stdout
    lines = out.splitlines()
    assert lines[0].startswith("kernels 5, span 0.4 ms, idle in gaps >= 15.0 us: 0.1 ms"), lines[0]
    assert any("hipMemcpy" in l and "k_fluxreg<1" in l and " 1 x" in l and "0.06 ms" in l for l in lines), out
    assert any("hipMemcpy" in l and "k_interp" in l and "0.03 ms" in l for l in lines), out  # (the last blocking call within 200 us of the launch)
    assert not any("teardown" in l or "setup_kernel" in l for l in lines), out
