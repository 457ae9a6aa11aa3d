import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def sg():
    import streamgen
    streamgen.build()
    return streamgen


@pytest.fixture(scope="session")
def H():
    """The product package; the shared library must already be built (no build on the GPU box needed:
    the in-tree .so travels with the snapshot).  Never falls back to anything else."""
    import h264decode_amd
    if not os.path.exists(os.path.join(ROOT, "h264decode_amd", "libh264mi.so")):
        h264decode_amd.build()
    h264decode_amd.load()
    return h264decode_amd


REAL_MP4 = "/opt/conda/lib/python3.9/site-packages/imageio/resources/images/realshort.mp4"


@pytest.fixture(scope="session")
def real_stream():
    if not os.path.exists(REAL_MP4):
        pytest.skip("third-party sample MP4 not present on this machine")
    from mp4util import mp4_to_annexb
    return mp4_to_annexb(open(REAL_MP4, "rb").read())


# the parity matrix shared by CPU (oracle vs generator) and GPU (product vs oracle + generator) tests
BASE = dict(width=176, height=144, frames=4, idr_period=0)
# Set when libh264mi decodes B slices; until then the GPU tests assert the documented refusal (H264MI_EUNSUPPORTED = -3).
PRODUCT_DECODES_B = True

MATRIX = {
    # Slice groups (FMO, 8.2.2; h264/slice.go:134-158, :457-552) and arbitrary slice order: all seven map types, one or two slices per
    # group, the slices of a picture in shuffled order, slice-edge deblocking off (idc 2), map units of two rows (interlace SPS)
    "fmo_interleaved": dict(BASE, profile_idc=66, cabac=0, slice_groups=3, fmo_type=0, intra_in_p_permille=150, seed=91),
    "fmo_dispersed_aso": dict(BASE, profile_idc=66, cabac=0, slice_groups=4, fmo_type=1, slices=2, aso=1, intra_in_p_permille=150, seed=92),
    "fmo_foreground": dict(BASE, profile_idc=66, cabac=0, slice_groups=3, fmo_type=2, aso=1, sub8x8_permille=300, seed=93),
    "fmo_boxout": dict(BASE, frames=6, profile_idc=66, cabac=0, slice_groups=2, fmo_type=3, aso=1, intra_in_p_permille=150, seed=94),
    "fmo_raster": dict(BASE, frames=6, profile_idc=66, cabac=0, slice_groups=2, fmo_type=4, slices=2, seed=95),
    "fmo_wipe": dict(BASE, frames=6, profile_idc=66, cabac=0, slice_groups=2, fmo_type=5, aso=1, num_ref_frames=2, seed=96),
    "fmo_explicit": dict(BASE, profile_idc=66, cabac=0, slice_groups=5, fmo_type=6, slices=2, aso=1, deblock_idc=2, pcm_permille=30, intra_in_p_permille=200, seed=97),
    "fmo_interlace_sps": dict(width=176, height=160, frames=4, idr_period=0, profile_idc=66, cabac=0, interlace_sps=1, slice_groups=3, fmo_type=1, seed=98),
    "fmo_explicit_cabac_b": dict(BASE, frames=7, profile_idc=77, cabac=1, slice_groups=3, fmo_type=6, aso=1, bframes=2, num_ref_frames=3, seed=99),
    # found by tools/param_sweep.py --extreme: macroblocks far beyond the 3200 bits A.3.1 allows a macroblock_layer() (QP 1 on loud noise: 8000 and more;
    # a conforming encoder would send I_PCM) must not run off the entropy kernels' bit window -- CAVLC and CABAC, escape-coded levels throughout
    "oversized_mbs_cavlc": dict(BASE, frames=3, profile_idc=66, cabac=0, qp=1, noise=100, qp_jitter=5, seed=101),
    "oversized_mbs_cabac_8x8": dict(BASE, frames=3, profile_idc=100, cabac=1, transform8x8=1, qp=0, noise=100, chroma_qp_offset=-12, seed=102),
    "aso_only": dict(BASE, profile_idc=66, cabac=0, slices=4, aso=1, deblock_idc=2, intra_in_p_permille=150, seed=90),
    # found by tools/param_sweep.py: POC type 2, and a picture with frame_num 1 that carries memory management operation 5 -- the next picture has
    # frame_num 1 again, nothing in 7.4.1.2.4 separates the two, and with foreground slice groups the slice of macroblock 0 is not the first one
    # the same trap with POC type 0 (operation 5 in the picture with frame_num 1 / POC 2: the next picture has both again) and an evolving map whose first
    # macroblock moves: only slice_group_change_cycle and the marking script tell the two pictures apart
    "fmo_boxout_mmco5_equal_headers": dict(width=92, height=128, frames=4, profile_idc=77, seed=252101, qp=31, qp_jitter=2, idr_period=0, slices=2, num_ref_frames=3,
                                           alpha_off_div2=-3, beta_off_div2=2, chroma_qp_offset=-1, skip_permille=500, sub8x8_permille=400, noise=2, motion_x4=1, motion_y4=-14,
                                           cabac=1, cabac_init_idc=1, weighted_pred=2, mmco=1, slice_groups=2, fmo_type=3),
    "fmo_mmco5_equal_frame_num": dict(width=156, height=48, frames=6, profile_idc=77, seed=301113, qp=26, qp_jitter=2, idr_period=5, slices=2, num_ref_frames=2,
                                      constrained_intra=1, sub8x8_permille=400, noise=20, long_start_code=0, motion_x4=12, motion_y4=2, cabac=1, cabac_init_idc=-1,
                                      weighted_pred=2, poc_type=2, mmco=1, idr_long_term=1, slice_groups=3, fmo_type=2, alpha_off_div2=-1, beta_off_div2=3,
                                      skip_permille=100),
    "cavlc_I": dict(width=64, height=48, frames=2, idr_period=1, profile_idc=66, cabac=0),
    "cabac_I": dict(width=64, height=48, frames=2, idr_period=1, profile_idc=77, cabac=1),
    "cavlc_IPP": dict(BASE, profile_idc=66, cabac=0, qp=24),
    "cabac_IPP": dict(BASE, profile_idc=77, cabac=1, qp=24),
    "cabac_lowqp": dict(BASE, frames=3, profile_idc=77, cabac=1, qp=8, noise=30),
    "cavlc_lowqp": dict(BASE, frames=3, profile_idc=66, cabac=0, qp=8, noise=30),
    "cabac_highqp": dict(BASE, profile_idc=77, cabac=1, qp=45),
    "cavlc_highqp": dict(BASE, profile_idc=66, cabac=0, qp=40),
    "high8x8_cabac": dict(BASE, profile_idc=100, cabac=1, transform8x8=1, qp=26),
    "high8x8_cavlc": dict(BASE, profile_idc=100, cabac=0, transform8x8=1, qp=26),
    "scaling_matrix": dict(BASE, profile_idc=100, cabac=1, transform8x8=1, qp=26, scaling_matrix=1),
    "slices_idc_cycle": dict(BASE, profile_idc=77, cabac=1, slices=3, cabac_init_idc=-1),
    "slices_dbf2": dict(BASE, profile_idc=66, cabac=0, slices=4, deblock_idc=2),
    "multiref_cabac": dict(BASE, frames=6, profile_idc=77, cabac=1, num_ref_frames=3, qp=30),
    "multiref_cavlc": dict(BASE, frames=6, profile_idc=66, cabac=0, num_ref_frames=4, qp=30),
    "pcm_qpjitter_cabac": dict(BASE, profile_idc=77, cabac=1, pcm_permille=60, qp_jitter=6),
    "pcm_qpjitter_cavlc": dict(BASE, profile_idc=66, cabac=0, pcm_permille=60, qp_jitter=6),
    "weighted_pred": dict(BASE, frames=5, profile_idc=77, cabac=1, weighted_pred=1, num_ref_frames=2),
    # both weight denominators 7: the inferred default weight of an entry without a flag is 128, outside the coded range
    "weighted_pred_denom7": dict(BASE, frames=5, profile_idc=77, cabac=0, weighted_pred=2, num_ref_frames=3, seed=61),
    # scene motion that is not a whole number of samples per frame: fractional vectors are the rule (every class of 8.4.2.2.1)
    "frac_motion_cabac": dict(BASE, frames=5, profile_idc=77, cabac=1, motion_x4=11, motion_y4=-6, seed=63),
    "frac_motion_sub8x8": dict(BASE, frames=5, profile_idc=66, cabac=0, sub8x8_permille=500, motion_x4=9, motion_y4=-7, num_ref_frames=2, seed=64),
    "frac_motion_high8x8": dict(BASE, frames=5, profile_idc=100, cabac=1, transform8x8=1, motion_x4=2, motion_y4=3, qp=24, seed=66),
    "constrained_intra": dict(BASE, profile_idc=77, cabac=1, constrained_intra=1, intra_in_p_permille=300),
    "no_deblock": dict(BASE, profile_idc=77, cabac=1, deblock_idc=1),
    "dbf_offsets_cqp": dict(BASE, profile_idc=77, cabac=1, alpha_off_div2=3, beta_off_div2=-2, chroma_qp_offset=4),
    "sub8x8_heavy": dict(BASE, profile_idc=77, cabac=1, sub8x8_permille=600, skip_permille=100, qp=24),
    "sub8x8_cavlc_poc2": dict(BASE, profile_idc=66, cabac=0, sub8x8_permille=600, skip_permille=100, qp=24, poc_type=2),
    "crop_3byte_sc": dict(width=180, height=100, frames=3, idr_period=0, profile_idc=100, cabac=1, transform8x8=1, long_start_code=0, slices=2),
    "idc1": dict(BASE, profile_idc=77, cabac=1, cabac_init_idc=1),
    "idc2": dict(BASE, profile_idc=77, cabac=1, cabac_init_idc=2),
    "gop3": dict(BASE, frames=7, idr_period=3, profile_idc=77, cabac=1),
    # picture management (8.2.1, 8.2.4.3, 8.2.5.4): list modification, MMCO 1..6, long-term pictures, POC type 1,
    # non-reference pictures, slice_qp_delta != 0
    "rplm_cabac": dict(BASE, frames=8, profile_idc=77, cabac=1, num_ref_frames=4, rplm=1, qp=30, seed=31),
    "rplm_cavlc": dict(BASE, frames=8, profile_idc=66, cabac=0, num_ref_frames=3, rplm=1, qp=30, seed=32),
    "mmco_cabac": dict(BASE, frames=24, profile_idc=77, cabac=1, num_ref_frames=4, rplm=1, mmco=1, qp=32, seed=33),
    "mmco_cavlc": dict(BASE, frames=24, profile_idc=66, cabac=0, num_ref_frames=3, rplm=1, mmco=1, qp=32, seed=34),
    "mmco_idr_lt": dict(BASE, frames=20, idr_period=10, profile_idc=77, cabac=1, num_ref_frames=3, rplm=1, mmco=1, idr_long_term=1, qp=32, seed=35),
    "poc1_nonref": dict(BASE, frames=9, profile_idc=77, cabac=1, num_ref_frames=2, poc_type=1, nonref_period=3, qp=30, seed=36),
    "poc2_nonref_cavlc": dict(BASE, frames=9, profile_idc=66, cabac=0, num_ref_frames=2, poc_type=2, nonref_period=2, qp=30, seed=37),
    "poc1_mmco": dict(BASE, frames=20, profile_idc=77, cabac=1, num_ref_frames=4, poc_type=1, mmco=1, rplm=1, nonref_period=4, qp=32, seed=38),
    "slice_qp_delta": dict(BASE, frames=5, profile_idc=77, cabac=1, slices=3, slice_qp_delta=5, qp=27, seed=39),
    "slice_qp_delta_cavlc": dict(BASE, frames=5, profile_idc=66, cabac=0, slices=2, slice_qp_delta=7, qp_jitter=3, qp=30, seed=40),
    # frame_num gaps (8.2.5.2) with gaps_in_frame_num_value_allowed_flag: "non-existing" frames go through the sliding window, the
    # lists are re-ordered around them; POC types 0, 2 and 1 (FrameNumOffset must survive the skipped values)
    "fn_gaps_cabac": dict(BASE, frames=14, profile_idc=77, cabac=1, num_ref_frames=4, fn_gap_period=3, fn_gap_declared=1, qp=30, seed=71),
    "fn_gaps_cavlc_poc2": dict(BASE, frames=14, idr_period=7, profile_idc=66, cabac=0, num_ref_frames=3, fn_gap_period=3, fn_gap_declared=1, qp=30, poc_type=2, seed=72),
    "fn_gaps_poc1_nonref": dict(BASE, frames=12, profile_idc=77, cabac=1, num_ref_frames=3, fn_gap_period=4, fn_gap_declared=1, qp=30, poc_type=1, nonref_period=3, seed=73),
    # frame_mbs_only_flag = 0 with mb_adaptive_frame_field_flag = 0 and field_pic_flag = 0 everywhere (first step):
    # the SPS / slice header syntax of a PAFF-capable stream whose pictures are all frames; crop units of four rows
    "interlace_sps_cabac": dict(width=176, height=120, frames=5, idr_period=0, profile_idc=77, cabac=1, interlace_sps=1, num_ref_frames=2, seed=81),
    "interlace_sps_cavlc_b": dict(width=176, height=128, frames=8, idr_period=0, profile_idc=77, cabac=0, interlace_sps=1, bframes=2, num_ref_frames=3,
                                  direct_temporal=1, seed=82),
    # B pictures: IBP / IBBP / IBBBP coding orders, spatial and temporal direct, B_Skip / B_Direct / all 22
    # inter mb_types and 13 sub_mb_types, default / explicit / implicit bi-prediction weights, list 1
    "b_ibp_cabac": dict(BASE, frames=9, profile_idc=77, cabac=1, bframes=1, num_ref_frames=2, bskip_permille=200, seed=41),
    "b_ibp_cavlc": dict(BASE, frames=9, profile_idc=77, cabac=0, bframes=1, num_ref_frames=2, bskip_permille=200, seed=42),
    "b_ibbp_cabac": dict(BASE, frames=10, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, bskip_permille=200, sub8x8_permille=300, seed=43),
    "b_ibbp_cavlc": dict(BASE, frames=10, profile_idc=77, cabac=0, bframes=2, num_ref_frames=3, bskip_permille=200, sub8x8_permille=300, seed=44),
    "b_temporal_cabac": dict(BASE, frames=13, profile_idc=77, cabac=1, bframes=2, num_ref_frames=2, direct_temporal=1, bskip_permille=300, seed=45),
    "b_temporal_cavlc": dict(BASE, frames=13, profile_idc=77, cabac=0, bframes=3, num_ref_frames=3, direct_temporal=1, bskip_permille=300, seed=46),
    "b_wp_explicit": dict(BASE, frames=10, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, weighted_bipred=1, weighted_pred=1, seed=47),
    "b_wp_explicit_denom7": dict(BASE, frames=10, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, weighted_bipred=1, weighted_pred=2, seed=62),
    "b_frac_motion": dict(BASE, frames=10, profile_idc=77, cabac=1, bframes=2, num_ref_frames=3, motion_x4=10, motion_y4=-5, weighted_bipred=2, seed=65),
    "b_wp_implicit": dict(BASE, frames=10, profile_idc=77, cabac=0, bframes=2, num_ref_frames=4, weighted_bipred=2, direct_temporal=1, seed=48),
    "b_high8x8_slices": dict(BASE, frames=10, profile_idc=100, cabac=1, transform8x8=1, bframes=2, num_ref_frames=2, slices=3, sub8x8_permille=400,
                             cabac_init_idc=-1, seed=49),
    # B pyramids: the middle B picture of a group is a reference picture (nal_ref_idc 2) -- B pictures in both lists and as
    # the co-located picture (its blocks may use list 1 only), three levels of entropy launches on the GPU
    "b_pyramid_cabac": dict(BASE, frames=13, profile_idc=77, cabac=1, bframes=3, b_pyramid=1, direct_temporal=1, bskip_permille=300, sub8x8_permille=200, seed=51),
    "b_pyramid_cavlc": dict(BASE, frames=13, profile_idc=77, cabac=0, bframes=3, b_pyramid=1, bskip_permille=300, sub8x8_permille=200, seed=52),
    "b_pyramid_implicit": dict(BASE, frames=14, profile_idc=100, cabac=1, transform8x8=1, bframes=2, b_pyramid=1, direct_temporal=1, weighted_bipred=2, slices=2,
                               idr_period=7, seed=53),
    "b_gop_intra_pcm": dict(BASE, frames=16, idr_period=8, profile_idc=77, cabac=1, bframes=3, num_ref_frames=3, intra_in_p_permille=200,
                            pcm_permille=100, qp_jitter=5, rplm=1, seed=50),
}


# Field pictures (PAFF: every frame coded as two fields, or picture-adaptively as a frame or two fields).  The product decodes them (round 4):
# the GPU parity tests run FULL_MATRIX = MATRIX + FIELD_MATRIX through every kernel-plan variant.  They are kept apart from MATRIX only because
# a frame of these streams is two pictures (two access units, two decoder "pictures per batch"), which the CPU tests over MATRIX that count
# access units or pictures would have to special-case.  CAVLC here; the same recipes with CABAC: FIELD_CABAC_MATRIX below.
FIELD_BASE = dict(width=176, height=128, frames=5, idr_period=0, profile_idc=77, cabac=0, field_pics=1)
FIELD_MATRIX = {
    "field_IP": dict(FIELD_BASE, num_ref_frames=2, seed=301),
    "field_intra_only": dict(FIELD_BASE, idr_period=1, frames=3, pcm_permille=30, seed=302),
    "field_bottom_first": dict(FIELD_BASE, field_pics=2, num_ref_frames=3, sub8x8_permille=400, seed=303),
    "field_refs4_qpel": dict(FIELD_BASE, num_ref_frames=4, frames=7, motion_x4=5, motion_y4=3, sub8x8_permille=300, seed=304),
    "field_high_8x8": dict(FIELD_BASE, profile_idc=100, transform8x8=1, scaling_matrix=1, num_ref_frames=2, intra_in_p_permille=200, seed=305),
    "field_slices3_idc2": dict(FIELD_BASE, slices=3, deblock_idc=2, alpha_off_div2=2, beta_off_div2=-1, num_ref_frames=2, seed=306),
    "field_wp_poc1": dict(FIELD_BASE, weighted_pred=1, poc_type=1, num_ref_frames=3, idr_period=3, frames=7, seed=307),
    "field_poc2_qpjitter": dict(FIELD_BASE, poc_type=2, qp_jitter=6, slice_qp_delta=2, skip_permille=300, num_ref_frames=2, seed=308),
    "field_mixed_paff": dict(FIELD_BASE, field_pics=3, frames=10, idr_period=6, num_ref_frames=3, sub8x8_permille=300, seed=310),
    "field_mixed_nonref_high": dict(FIELD_BASE, field_pics=3, frames=9, nonref_period=3, num_ref_frames=2, profile_idc=100, transform8x8=1, weighted_pred=2, seed=311),
    "field_nonref_pairs": dict(FIELD_BASE, frames=7, nonref_period=2, num_ref_frames=2, poc_type=2, seed=312),
    "field_b_spatial": dict(FIELD_BASE, frames=9, bframes=2, num_ref_frames=2, bskip_permille=250, sub8x8_permille=300, seed=313),
    "field_b_bottom_first_wp": dict(FIELD_BASE, field_pics=2, frames=8, bframes=1, num_ref_frames=3, weighted_bipred=1, intra_in_p_permille=100, seed=314),
    "field_b_implicit_high": dict(FIELD_BASE, frames=9, bframes=3, num_ref_frames=2, weighted_bipred=2, profile_idc=100, transform8x8=1, idr_period=8, slices=2, seed=315),
    "field_b_temporal": dict(FIELD_BASE, frames=9, bframes=2, direct_temporal=1, num_ref_frames=3, bskip_permille=400, seed=316),
    "field_b_temporal_implicit_bff": dict(FIELD_BASE, field_pics=2, frames=9, bframes=3, direct_temporal=1, weighted_bipred=2, num_ref_frames=2, bskip_permille=300, idr_period=8, seed=317),
    "field_rplm": dict(FIELD_BASE, frames=8, rplm=1, num_ref_frames=3, seed=318),
    "field_rplm_mixed_nonref": dict(FIELD_BASE, field_pics=3, frames=10, rplm=1, nonref_period=4, num_ref_frames=4, idr_period=7, seed=319),
    "field_mmco1": dict(FIELD_BASE, frames=10, mmco=1, num_ref_frames=3, sub8x8_permille=200, seed=320),
    "field_mmco1_rplm_mixed": dict(FIELD_BASE, field_pics=3, frames=12, mmco=1, rplm=1, num_ref_frames=2, idr_period=8, seed=321),
    "field_fmo_dispersed": dict(FIELD_BASE, frames=4, slice_groups=3, fmo_type=1, num_ref_frames=2, seed=322),
    "field_fmo_boxout_mixed_aso": dict(FIELD_BASE, field_pics=3, frames=8, slice_groups=2, fmo_type=3, slices=2, aso=1, num_ref_frames=2, seed=323),
    "field_fmo_explicit_bff": dict(FIELD_BASE, field_pics=2, frames=4, slice_groups=4, fmo_type=6, num_ref_frames=2, seed=324),
    "field_cropped": dict(FIELD_BASE, width=170, height=124, num_ref_frames=2, constrained_intra=1, intra_in_p_permille=150, seed=309),
}


# Picture order counts with a bottom field that is not at the top field's count (bottom_field_pic_order_in_frame_present_flag = 1:
# delta_pic_order_cnt_bottom / delta_pic_order_cnt[1]).  Part of MATRIX (below): every GPU variant, the golden MD5s and the CPU tests see them;
# on the CPU the product's HOST side (built against the null device of tools/hoststub) must arrive at the same PicOrderCnt for every picture
# -- tests/test_host_picture_management.py.
POC_MATRIX = {
    "poc_bottom_later": dict(BASE, frames=8, profile_idc=77, cabac=1, num_ref_frames=2, poc_bottom_delta=1, seed=401),
    "poc_bottom_first": dict(BASE, frames=8, profile_idc=77, cabac=0, num_ref_frames=2, poc_bottom_delta=-1, idr_period=5, seed=402),
    "poc_bottom_first_b_temporal": dict(BASE, frames=10, profile_idc=77, cabac=1, num_ref_frames=3, poc_bottom_delta=-1, bframes=2, direct_temporal=1, weighted_bipred=2, seed=403),
    "poc_bottom_first_poc1_mmco": dict(BASE, frames=16, profile_idc=77, cabac=0, num_ref_frames=3, poc_bottom_delta=-1, poc_type=1, mmco=1, seed=404),
    "poc_bottom_first_mmco5": dict(BASE, frames=24, profile_idc=77, cabac=1, num_ref_frames=3, poc_bottom_delta=-1, mmco=1, seed=21),
    "poc_bottom_later_interlace_sps": dict(BASE, height=128, frames=6, profile_idc=100, cabac=1, transform8x8=1, num_ref_frames=2, poc_bottom_delta=2, interlace_sps=1, seed=405),
}

MATRIX.update(POC_MATRIX)

# Monochrome (chroma_format_idc 0; h264/sps.go:226-243 ChromaFormat, h264/slice.go:179-219 SubWidthC / SubHeightC; round 5, SURVEY 8f rank 4 first step): High
# profile streams without any chroma syntax -- no intra_chroma_pred_mode, coded_block_pattern by the ChromaArrayType 0 column of Table 9-4
# (h264/bit_reader.go:118-135; CABAC: the prefix only), 256 samples per I_PCM macroblock, no chroma weights.  Decoded pictures carry chroma planes of 128.
MONO_MATRIX = {
    "mono_cabac_8x8_pcm_wp": dict(BASE, profile_idc=100, mono=1, cabac=1, transform8x8=1, pcm_permille=30, weighted_pred=1, num_ref_frames=2, frames=5, seed=501),
    "mono_cavlc_slices_pcm": dict(BASE, profile_idc=100, mono=1, cabac=0, pcm_permille=30, num_ref_frames=2, slices=2, deblock_idc=0, intra_in_p_permille=200, seed=502),
    "mono_b_explicit_cabac": dict(BASE, profile_idc=100, mono=1, cabac=1, bframes=2, weighted_bipred=1, num_ref_frames=3, frames=8, bskip_permille=300, seed=503),
    "mono_b_implicit_cavlc": dict(BASE, profile_idc=100, mono=1, cabac=0, bframes=1, weighted_bipred=2, frames=7, intra_in_p_permille=300, transform8x8=1, seed=504),
    "mono_intra_only_scaling": dict(BASE, profile_idc=100, mono=1, cabac=1, idr_period=1, frames=3, scaling_matrix=1, transform8x8=1, pcm_permille=20, seed=505),
    "mono_cropped_cavlc_qpel": dict(BASE, width=170, height=138, profile_idc=100, mono=1, cabac=0, motion_x4=5, motion_y4=-3, sub8x8_permille=300, num_ref_frames=2, seed=506),
}
MATRIX.update(MONO_MATRIX)
FULL_MATRIX = dict(MATRIX, **FIELD_MATRIX)

# The same 24 field recipes with CABAC (round 5).  Field-coded blocks have significance contexts of their own (ctxIdx 277-398, 436-459, the field column of
# Table 9-43); their initialisation values in the three *_cabac_mn.* copies are UNPINNED (entered without the standard at hand, no third-party stream), so
# the product decodes such pictures only with h264mi_config.allow_unpinned_field_cabac = 1.  What these cases pin: generator == oracle == GPU, i.e. the
# mechanism (context offsets, field map, field scans under CABAC) -- not the values.
FIELD_CABAC_MATRIX = {name + "_cabac": dict(kw, cabac=1) for name, kw in FIELD_MATRIX.items()}


def pictures_of(kw):
    """Pictures (access units) of a matrix stream: a frame coded as two fields is two of them.  What a decoder's max_frames_per_batch counts."""
    return kw["frames"] * (2 if kw.get("field_pics") else 1)


def with_extension_nals(stream, seed=1):
    """The Annex-B stream with the NAL units of an SVC / MVC / 3D-AVC stream around its own (Annex G / H / J: prefix NAL unit 14 in front of every
    slice, subset SPS 15 behind every SPS, coded slice extension 20 / 21 behind every slice), an SEI and filler data.  A decoder of the base layer / base
    view ignores all of them (h264/nalUnit.go:39-71 parses their header extension; h264/server.go:147-164 looks at types 1, 5, 7, 8 only)."""
    import random
    rnd = random.Random(seed)
    # split at start codes (3- or 4-byte), keep each unit with its start code
    units, i, n = [], 0, len(stream)
    starts = []
    while i + 3 <= n:
        if stream[i] == 0 and stream[i + 1] == 0 and stream[i + 2] == 1:
            starts.append(i - 1 if i > 0 and stream[i - 1] == 0 else i)
            i += 3
        else:
            i += 1
    starts.append(n)
    for a, b in zip(starts, starts[1:]):
        units.append(bytes(stream[a:b]))

    def ext(nal_type, ref_idc, n_bytes, svc=True):
        body = bytes(rnd.choice((0x55, 0xAA, 0x7F, 0x33, 0x81)) for _ in range(n_bytes))  # (no 00 00 0x inside)
        hdr = bytes([(ref_idc << 5) | nal_type])
        if nal_type in (14, 20, 21):
            hdr += bytes([0x80 | rnd.randrange(64) if svc else rnd.randrange(64), 0x80 | rnd.randrange(128), 0x07 | (rnd.randrange(32) << 3)])
        return b"\x00\x00\x01" + hdr + body + b"\x80"
    out = []
    for u in units:
        k = 4 if u[:4] == b"\x00\x00\x00\x01" else 3
        t = u[k] & 31
        if t in (1, 5):
            out.append(ext(14, (u[k] >> 5) & 3, 0))
        out.append(u)
        if t == 7:
            out.append(ext(15, 3, 24))
        if t in (1, 5):
            out.append(ext(20, (u[k] >> 5) & 3, rnd.randrange(8, 200), svc=rnd.random() < 0.5))
            if rnd.random() < 0.3:
                out.append(ext(21, 0, rnd.randrange(8, 60)))
            if rnd.random() < 0.3:
                out.append(b"\x00\x00\x01\x06\x05\x04\x55\x55\x55\x55\x80")  # an SEI
            if rnd.random() < 0.3:
                out.append(b"\x00\x00\x01\x0c" + b"\xff" * rnd.randrange(1, 40) + b"\x80")  # filler data
    return b"".join(out)
