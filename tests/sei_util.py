"""Decoded-picture-hash SEI messages (H.265 D.2.19 / D.3.19; parsed by the reference in sei.cc:32-62) for generated streams: the
stream is decoded once by the REFERENCE decoder, every picture is hashed with the oracle's restatement of sei.cc:161-274, and a
suffix SEI NAL unit carrying the hash is placed behind the last slice segment of each picture.  Test infrastructure."""
import numpy as np

import hash_util

MD5, CRC, CHECKSUM = 0, 1, 2


def split_nals(data):
    """-> list of NAL units (each without its start code)"""
    out, i, n = [], 0, len(data)
    starts = []
    while True:
        j = data.find(b"\x00\x00\x01", i)
        if j < 0:
            break
        starts.append(j + 3)
        i = j + 3
    for k, s in enumerate(starts):
        e = starts[k + 1] - 3 if k + 1 < len(starts) else n
        while e > s and data[e - 1] == 0 and k + 1 < len(starts):     # zero_byte / trailing_zero_8bits in front of the next start code
            e -= 1
        out.append(data[s:e])
    return out


def _escape(payload):
    """emulation prevention (7.4.2): 00 00 0x -> 00 00 03 0x for x <= 3"""
    out, zeros = bytearray(), 0
    for b in payload:
        if zeros >= 2 and b <= 3:
            out.append(3)
            zeros = 0
        out.append(b)
        zeros = zeros + 1 if b == 0 else 0
    return bytes(out)


def hash_sei_nal(planes, bit_depth, hash_type, olib, corrupt=False):
    body = bytearray([hash_type])
    for c, pl in enumerate(planes):
        v = hash_util.oracle_hash(olib, pl, bit_depth, hash_type)
        if hash_type == MD5:
            v = bytearray(v)
            if corrupt and c == len(planes) - 1:
                v[5] ^= 0x10
            body += v
        elif hash_type == CRC:
            body += int((v ^ (1 if corrupt and c == 0 else 0)) & 0xFFFF).to_bytes(2, "big")
        else:
            body += int((v + (1 if corrupt and c == 1 % len(planes) else 0)) & 0xFFFFFFFF).to_bytes(4, "big")
    assert len(body) < 255
    rbsp = bytes([132, len(body)]) + bytes(body) + b"\x80"           # payload_type, payload_size, payload, rbsp_trailing_bits
    return bytes([40 << 1, 1]) + _escape(rbsp)                        # SUFFIX_SEI_NUT, nuh_layer_id 0, temporal_id_plus1 1


def add_hash_seis(data, pictures, bit_depth, hash_type, olib, corrupt_picture=None):
    """data: the stream; pictures: every picture's planes in DECODE order (= output order in the generated streams)"""
    nals = split_nals(data)
    out, pic = [], -1
    for k, nal in enumerate(nals):
        t = (nal[0] >> 1) & 0x3F
        vcl = t < 32
        if vcl and (nal[2] & 0x80):                  # first_slice_segment_in_pic_flag: a new picture begins here
            if pic >= 0:
                out.append(hash_sei_nal(pictures[pic], bit_depth, hash_type, olib, corrupt_picture == pic))
            pic += 1
        out.append(nal)
    if pic >= 0:
        out.append(hash_sei_nal(pictures[pic], bit_depth, hash_type, olib, corrupt_picture == pic))
    assert pic + 1 == len(pictures)
    return b"".join(b"\x00\x00\x00\x01" + n for n in out)
