"""Per-kernel ISA statistics of a hipcc -save-temps device assembly file: register counts, scratch, MFMA / LDS-transpose / convert
instruction counts.  Usage: python tools/isa_stats.py <file.s> <kernel-name-substring> [...]"""
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    for name in sys.argv[2:]:
        for m in re.finditer(r'\n(_Z\d+%s\w*):[^\n]*\n' % re.escape(name), s):
            st = m.end()
            en = s.find('.end_amdhsa_kernel', st)
            body = s[st:en]

            def cnt(pat):
                return len(re.findall(pat, body))
            print(m.group(1)[:60])
            print('   mfma_f32x4 %d  mfma_bf16 %d  ds_read_tr %d  cvt_pk_bf16 %d  scratch %d  flat %d' % (
                cnt('v_mfma_f32_16x16x4'), cnt('v_mfma_f32_16x16x32_bf16'), cnt('ds_read_b64_tr_b16'), cnt('v_cvt_pk_bf16'),
                cnt('scratch_'), cnt(r'\bflat_')))
            for k in ['next_free_vgpr', 'next_free_sgpr', 'accum_offset', 'private_segment_fixed_size', 'group_segment_fixed_size']:
                mm = re.search(r'\.amdhsa_%s (\S+)' % k, body)
                print('   ', k, mm.group(1) if mm else None)


if __name__ == '__main__':
    main()
