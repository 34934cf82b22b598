import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.getcwd()+"/tools"); sys.path.insert(0, os.getcwd())
from kbench import run
S = 255 << 8
for C, T in ((384, 2000), (128, 8000), (192, 8000)):
    run(C, T, 256, flags=S)
    for fl, nm in ((0, "full"), (1, "no epilogue"), (16, "no global loads"), (17, "no epi + no loads"), (2, "no MFMA"), (18, "no MFMA, no loads"), (3, "no MFMA no epi"), (25, "MFMA only")):
        us, tf, gb, k = run(C, T, 256, flags=S | fl)
        print(f"C={C:4d} T={T:6d} {k:24s} {nm:22s} {us:9.1f} us {tf:6.1f} TF/s {gb:7.1f} GB/s", flush=True)
