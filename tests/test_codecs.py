"""Wire codecs against the reference's own vectors (src/reference/webgpu/utils.test.ts:4-14). CPU only."""
import pytest

import webgpu_msm_bls12_377_amd as msm

ALEO_FIELD_MODULUS = 8444461749428370424248824938781546531375899335154063827935233455917409239041

TEST_DATA = [
    (0, [0, 0, 0, 0, 0, 0, 0, 0]),
    (1, [0, 0, 0, 0, 0, 0, 0, 1]),
    (33, [0, 0, 0, 0, 0, 0, 0, 33]),
    (4294967297, [0, 0, 0, 0, 0, 0, 1, 1]),
    (ALEO_FIELD_MODULUS, [313222494, 2586617174, 1622428958, 1547153409, 1504343806, 3489660929, 168919040, 1]),
    (115792089237316195423570985008687907853269984665640564039457584007913129639935, [4294967295] * 8),
    (6924886788847882060123066508223519077232160750698452411071850219367055984476, [256858326, 3006847798, 1208683936, 2370827163, 3854692792, 1079629005, 1919445418, 2787346268]),
    (60001509534603559531609739528203892656505753216962260608619555, [0, 9558, 3401397337, 1252835688, 2587670639, 1610789716, 3992821760, 136227]),
    (30000754767301779765804869764101946328252876608481130304309778, [0, 4779, 1700698668, 2773901492, 1293835319, 2952878506, 1996410880, 68114]),
]


@pytest.mark.parametrize("value,words", TEST_DATA)
def test_bigIntToU32Array(value, words):
    assert msm.bigIntToU32Array(value) == words
    assert msm.bigIntsToU32Array([value]) == words


@pytest.mark.parametrize("value,words", TEST_DATA)
def test_u32ArrayToBigInts(value, words):
    assert msm.u32ArrayToBigInts(words) == [value]


def test_buffer_le_round_trip():
    vals = [v for v, _ in TEST_DATA]
    buf = msm.bigIntsToBufferLE(vals, 256)
    assert len(buf) == 32 * len(vals)
    assert buf[:32] == bytes(32) and buf[32] == 1
    keep = bytes(buf)
    assert msm.readBigIntsFromBufferLE(buf, 256) == vals
    assert buf == keep  # the input is not reversed in place (unlike src/reference/webgpu/utils.ts:78-79)
    assert msm.readBigIntsFromBufferLE(msm.bigIntsToBufferLE(vals, 384), 384) == vals


def test_point_and_scalar_forms():
    """The three input forms of compute_msm (submission.ts:86-87) encode to the same buffers."""
    pts = [{"x": 5, "y": 7, "z": 1}, {"x": 2**376 + 3, "y": 11, "z": 1}]
    buf = msm.points_to_buffer(pts)
    assert len(buf) == 192 and buf[:48] == (5).to_bytes(48, "little") and buf[48:96] == (7).to_bytes(48, "little")
    u32pts = [{"x": msm.bigIntToU32Array(p["x"], 384), "y": msm.bigIntToU32Array(p["y"], 384)} for p in pts]
    assert msm.points_to_buffer(u32pts) == buf
    assert msm.points_to_buffer(buf) == buf
    ks = [1, 2**252 + 9]
    sbuf = msm.scalars_to_buffer(ks)
    assert sbuf == msm.bigIntsToBufferLE(ks, 256)
    assert msm.scalars_to_buffer([msm.bigIntToU32Array(k) for k in ks]) == sbuf
