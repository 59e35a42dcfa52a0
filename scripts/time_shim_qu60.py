"""Times the MPAS shim modules at QU60 through fortran/harness/shim_harness (pools in ordinary, pageable memory, as a
framework allocates them): seaice_run_velocity_solver alone and with seaice_run_advection, with the shims' own timers;
plain, with SIH_SHIM_LAZY_DIAGNOSTICS and with SIH_SHIM_DEFERRED_DIAGNOSTICS."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpas_seaice_amd import cases, fortran_case, mesh as meshmod  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "QU60"
mesh = meshmod.make_named_mesh(name)
dom = cases.make_domain(mesh, name)
with tempfile.TemporaryDirectory() as d:
    case = os.path.join(d, "case.bin")
    fortran_case.write_pools(case, dom)
    print("case file MB", os.path.getsize(case) / 1e6, flush=True)
    for label, env_extra in (("plain", {}), ("lazy", {"SIH_SHIM_LAZY_DIAGNOSTICS": "1"}),
                             ("deferred", {"SIH_SHIM_DEFERRED_DIAGNOSTICS": "1"}),
                             ("lazy+deferred", {"SIH_SHIM_LAZY_DIAGNOSTICS": "1", "SIH_SHIM_DEFERRED_DIAGNOSTICS": "1"}),
                             ("keep state+deferred", {"SIH_SHIM_KEEP_STATE": "1", "SIH_SHIM_DEFERRED_DIAGNOSTICS": "1"})):
        env = dict(os.environ, **env_extra)
        env["LD_LIBRARY_PATH"] = "/opt/rocm/lib/llvm/lib:/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
        out = subprocess.run([os.path.join(ROOT, "fortran", "harness", "shim_harness"), case, os.path.join(d, "res.bin"),
                              "1", "time"], capture_output=True, text=True, timeout=500, env=env)
        print("====", label, "rc", out.returncode, flush=True)
        print("\n".join(l for l in out.stdout.splitlines() if "ms" in l or "ok" in l), flush=True)
        if out.returncode:
            print(out.stdout[-1500:], out.stderr[-1500:])
            sys.exit(1)
