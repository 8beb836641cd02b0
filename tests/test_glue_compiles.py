"""The HDK-side C++ bindings (hdk_amd/glue/) compile against the reference's own interface headers
where /root/reference exists (build container); skipped on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/omniscidb"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_hipmgr_glue_compiles_against_reference_gpumgr(tmp_path):
    src = tmp_path / "glue_check.cpp"
    src.write_text('#include "HipMgr.h"\n#include "HipRuntimeOnDevice.h"\n'
                   "// instantiates every override: HipMgr must not be abstract\n"
                   "GpuMgr* make() { return new hip_mgr::HipMgr(); }\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", REF, "-I",
                           os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "hdk_amd", "glue"), str(src)])


def test_public_header_is_plain_c(tmp_path):
    src = tmp_path / "c_check.c"
    src.write_text('#include "hdk_hip.h"\nint main(void) { return sizeof(hdk_hip_plan) > 0 ? 0 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Werror", "-pedantic", "-I",
                           os.path.join(ROOT, "include"), str(src)])
