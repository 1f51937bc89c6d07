#!/usr/bin/env python3
"""Static instruction counts per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only).

    python tools/isa_count.py dev.s [substring of the kernel name ...]
"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)^\s+s_endpgm', text, re.M | re.S):
        name, body = m.group(1), m.group(2)
        if pats and not any(p in name for p in pats):
            continue
        meta = re.search(r'\.vgpr_count:\s+(\d+)', text[text.find('.name:           ' + name):] or '')
        lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', text[text.find('.name:           ' + name):] or '')
        cnt = lambda rx: len(re.findall(rx, body, re.M))
        print('%-70s valu %4d salu %4d ds %3d vmem %3d dpp %3d nop %3d waitcnt %3d vgpr %s lds %s' % (
            name[:70], cnt(r'^\s+v_'), cnt(r'^\s+s_(?!nop|waitcnt)'), cnt(r'^\s+ds_'),
            cnt(r'^\s+(global|buffer|flat)_'), cnt(r'wave_sh[lr]|row_sh[lr]|row_bcast'), cnt(r'^\s+s_nop'),
            cnt(r'^\s+s_waitcnt'), meta.group(1) if meta else '?', lds.group(1) if lds else '?'))


if __name__ == '__main__':
    main()
