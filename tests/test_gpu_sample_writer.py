"""GPU: the SampleWriter mirror (src/rlUtil.h:43-171) as a statistical check -- the directions evalSample
draws are distributed like evalPdf says (chi-square of the lat-long histogram against the pdf integrated per
bin), for the GGX lobe and both rlDisney lobes."""
import numpy as np
import pytest

import rlshaders_amd as R
from rlshaders_amd.sample_writer import SampleWriter, write_ppm

pytestmark = pytest.mark.gpu

COUNT = 1 << 22


@pytest.mark.parametrize("rough,aniso,wo", [(0.5, 0.0, (0.0, 0.0, 1.0)), (0.6, 0.0, (0.6, 0.0, 0.8)),
                                           (0.7, 0.6, (0.5, 0.4, 0.6)), (0.9, 0.3, (0.9, 0.1, 0.3))])
def test_ggx_samples_follow_the_pdf(gpu, rough, aniso, wo, tmp_path):
    sw = SampleWriter(gpu, 64, 16)

    def make(n):
        WO, N, T = sw.frame(n, wo)
        return R.GgxSampler(gpu, WO, N, T, specColor=(1, 1, 1), ior=1.5, roughness=rough, anisotropic=aniso)

    r = sw.compare(make, COUNT)
    print("ggx", rough, aniso, wo, r)
    assert r["invalid"] == 0
    assert abs(r["pdf_mass"] + r["below_horizon"] / COUNT - 1.0) < 0.02 or r["pdf_mass"] <= 1.02
    assert abs(r["sampled_mass"] - r["pdf_mass"]) < 0.01          # same mass above the horizon
    assert r["covered_mass"] > 0.9 * r["pdf_mass"] and r["dof"] > 50
    assert r["chi2_per_dof"] < 1.5
    img = sw.writeRadiance(make)
    assert img.shape == (16, 64, 3) and np.isfinite(img).all() and (img >= 0).all()
    write_ppm(str(tmp_path / "radiance.ppm"), img)
    assert (tmp_path / "radiance.ppm").stat().st_size == 16 * 64 * 3 + len(b"P6\n64 16\n255\n")


@pytest.mark.parametrize("lobe", ["diffuse", "glossy"])
def test_disney_samples_follow_the_pdf(gpu, lobe):
    sw = SampleWriter(gpu, 64, 16)

    def make(n):
        WO, N, T = sw.frame(n, (0.5, 0.2, 0.8))
        d = R.DisneySampler(gpu, WO, N, T, base_color=(0.8, 0.6, 0.4), metallic=0.3, roughness=0.6, anisotropic=0.4,
                            clearcoat=0.0, specular=0.5)
        d.setSampleType(R.RLS_RAY_DIFFUSE if lobe == "diffuse" else R.RLS_RAY_GLOSSY)
        return d

    r = sw.compare(make, COUNT)
    print("disney", lobe, r)
    assert abs(r["sampled_mass"] - r["pdf_mass"]) < 0.01
    assert r["dof"] > 50 and r["chi2_per_dof"] < 1.5
