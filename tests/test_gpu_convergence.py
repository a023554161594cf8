"""The 16-bit training tier proven on a MODEL, not on a step (north_star "PSNR within 0.05 dB of reference"):
fresh students trained with the production step in the exact tier (the reference's fp32 arithmetic,
run_nerf_com_trainExpLater.py:916-931) and in the 16-bit tier (bf16 forward / dX, MX-fp8 x MX-fp4 weight gradients) from the
same start on the same frame and pixel sequence, scored on held-out frames in the exact tier.  Harness: tests/convergence.py;
long form (12,000 steps, both activation formats, three seeds each, the noise floor): tools/convergence.py -> profiles/r05j_convergence.txt."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEPS = 3000


@pytest.fixture(scope="module")
def result():
    import convergence as CV
    return CV.run(STEPS, [("f32", "f32", None, 100), ("f32_other_pixels", "f32", None, 101), ("bf16_fp4", "bf16", "fp4", 100)])


def test_students_learn_the_scene(result):
    """the harness trains: every variant ends far above the common start on the held-out frames, with finite losses"""
    for name, info in result["variants"].items():
        assert info["finite"] and info["last_loss"] < info["first_loss"], (name, info["first_loss"], info["last_loss"])
        for im in ("head", "com"):
            assert info["psnr_held_out"][im] > result["untrained"][im] + GAIN_DB, (name, im, info["psnr_held_out"], result["untrained"])


def test_16bit_tier_trains_as_good_a_model_as_the_exact_tier(result):
    """PSNR of the 16-bit-trained student on the held-out frames (rendered in the exact tier) against the f32-trained students'.
    Two trajectories that differ in rounding - or in the pixel seed - end at slightly different models: the second f32 run measures
    that spread.  Measured on 12,000-step runs with three seeds per 16-bit format (profiles/r05j_convergence.txt; LABNOTES.md 9.2): on
    the training frames the 16-bit models are 0.14-0.26 dB better on the head image (49 dB) and within 0.02 dB on the composite; on the
    held-out frames the pixel seed moves a run by +-0.5 dB and the 16-bit means are 0.07-0.14 dB below the f32 pair's - no systematic
    deficit.  Gates: the 16-bit model is not worse than the WORSE of the two f32 models by more than GATE_*_DB plus twice their spread."""
    v = result["variants"]
    for which, key in (("held-out", "psnr_held_out"), ("training frames", "psnr_train_frames")):
        for im, gate in (("head", GATE_HEAD_DB), ("com", GATE_COM_DB)):
            ref, other, got = (v[k][key][im] for k in ("f32", "f32_other_pixels", "bf16_fp4"))
            noise = abs(other - ref)
            print(f"{which} {im}: f32 {ref:.3f} dB, f32 (other pixels) {other:.3f} dB, 16-bit tier {got:.3f} dB "
                  f"(difference to their mean {got - 0.5 * (ref + other):+.3f}, spread of the two f32 runs {noise:.3f})")
            assert got > min(ref, other) - gate - 2.0 * noise, (which, im, ref, other, got)


def test_16bit_trained_model_holds_the_f16_inference_clause(result):
    """the model the 16-bit tier trained, rendered in the f16 INFERENCE tier: >= 49.4 dB against the exact tier on the full frame
    and on every 2,500-ray block (the north star's 0.05 dB at a 30 dB model, LABNOTES.md 3) - on TRAINED weights, configs[1]
    (head, 64 + 128), configs[2] (two fields) and the coarse renderer"""
    chk = result["variants"]["bf16_fp4"]["f16_inference_vs_f32"]
    for tag, c in chk.items():
        print(f"{tag}: f16 vs f32 on the 16-bit-trained student: {c['psnr_db']:.2f} dB, worst block {c['worst_block_db']:.2f} dB")
        assert c["finite"] and c["psnr_db"] >= 49.4 and c["worst_block_db"] >= 49.4, (tag, c)
    # the tier's accuracy guard (round 6) on the same weights, gated at what the clause implies for THIS model's own PSNR against the
    # scene's ground truth (measured on the held-out frames): accepted, with its margins printed
    g = result["variants"]["bf16_fp4"]["f16_accuracy_guard"]
    for im in ("head", "com"):
        q = g[im]
        print(f"accuracy guard, {im}: f16 vs f32 {q['psnr_db']:.2f} dB (worst frame {q['worst_block_db']:.2f}), model {q['model_psnr_db']:.2f} dB "
              f"-> gate {q['gate_db']:.2f} dB, margin {q['margin_db']:+.2f} dB")
    assert g["verdict"] == "accepted", g["verdict"]


# set from the measured runs (profiles/r05j_convergence.txt, r05j_convergence_test.txt; LABNOTES.md 9.2): the students start at 7.4 dB and
# reach 28-30 dB on both held-out images in 3,000 steps
GAIN_DB = 8.0
GATE_HEAD_DB = 0.5
GATE_COM_DB = 0.1
