"""CPU suite, part 5: include/mi_denoise.h is plain C (no C++, no torch types) and a C program
links against libmi_denoise.so -- the shape of the binding INTEGRATION.md describes."""
import os
import subprocess
import textwrap

from conftest import ROOT

SRC = textwrap.dedent(r'''
    #include <stdio.h>
    #include <string.h>
    #include "mi_denoise.h"

    int main(void)
    {
        /* the parameter blocks begin with the reference's push-constant layouts */
        mid_bilateral_params bp = {1920, 1080, 2.0f, 0.2f, 20, MID_LAYOUT_TEXTURE, MID_FMT_RGBA32F};
        mid_nlm_params np = {1920, 1080, 0.5f, -7, 7, -3, 3, MID_FMT_RGBA8};
        mid_normalize_params zp = {1920, 1080};
        if (sizeof(mid_weightinfo) != 32 || sizeof(mid_pixel) != 16 || sizeof zp != 8) return 10;
        if ((char *)&bp.radius - (char *)&bp != 16 || (char *)&np.search_lo - (char *)&np != 12) return 11;
        if (mid_version() != MID_VERSION) return 12;
        mid_ctx *ctx = NULL;
        int rc = mid_ctx_create(0, &ctx);
        if (rc == MID_OK) { mid_ctx_destroy(ctx); puts("ctx ok"); return 0; }      /* a GPU box */
        if (rc != MID_ERR_NO_DEVICE || ctx != NULL) return 13;
        if (!strstr(mid_last_error(), "no CPU path")) return 14;
        /* no context: every compute entry point refuses */
        if (mid_bilateral(NULL, &bp, NULL, NULL, NULL) != MID_ERR_INVALID) return 15;
        if (mid_nlm_accum(NULL, &np, NULL, NULL, NULL, NULL) != MID_ERR_INVALID) return 16;
        { mid_recording *rec = NULL;        /* recorded command sequences: no context, no recording */
          if (mid_record_begin(NULL, NULL) != MID_ERR_INVALID || mid_record_end(NULL, NULL, &rec) != MID_ERR_INVALID || rec != NULL) return 17;
          if (mid_recording_submit(NULL, NULL) != MID_ERR_INVALID || mid_recording_destroy(NULL) != MID_OK) return 18; }
        puts("no device, refused");
        return 0;
    }
''')


def test_header_is_c_and_links(tmp_path):
    c = tmp_path / "bind.c"
    c.write_text(SRC)
    exe = tmp_path / "bind"
    libdir = os.path.join(ROOT, "image_denoising_filter_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe),
                    "-L", libdir, "-lmi_denoise", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.strip() in ("ctx ok", "no device, refused")
