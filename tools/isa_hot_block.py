"""Instruction mix of the basic block with the most MFMAs of one kernel in a device assembly file (hipcc --cuda-device-only -S;
tools/build_variants.sh leaves /tmp/<source>_<TAG>.s):
    python tools/isa_hot_block.py /tmp/ea_attention_V4F2.s V4F2 [kernel-name substring, default attention_fwd_v3_kernelILi0ELb0E]"""
import sys,re,collections
s=open(sys.argv[1]).read()
kern=sys.argv[3] if len(sys.argv) > 3 else 'attention_fwd_v3_kernelILi0ELb0E'
m=re.search(r'\n(_ZN\S*' + kern + r'\S*?):[^\n]*\n(.*?)\.Lfunc_end', s, re.S)
body=m.group(2)
blocks=re.split(r'\n(?=\.LBB\d+_\d+:)', body)
best=max(blocks,key=lambda b:len(re.findall(r'v_mfma',b)))
c=collections.Counter(re.findall(r'^\s+((?:v|s|ds|global|buffer|scratch)_\w+)', best, re.M))
tot_v=sum(v for k,v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
print(sys.argv[2], 'hot block: mfma',sum(v for k,v in c.items() if k.startswith('v_mfma')),'valu',tot_v,'salu',sum(v for k,v in c.items() if k.startswith('s_')),'ds',sum(v for k,v in c.items() if k.startswith('ds_')))
print('  ',{k:v for k,v in sorted(c.items()) if k.startswith('v_') and not k.startswith('v_mfma')})
print('  ',{k:v for k,v in sorted(c.items()) if k.startswith('s_')})
k=re.search(r'\.amdhsa_kernel '+re.escape(m.group(1))+r'(.*?)\.end_amdhsa_kernel', s, re.S).group(1)
print('  ',re.findall(r'next_free_vgpr \d+|accum_offset \d+|private_segment_fixed_size \d+', k))
