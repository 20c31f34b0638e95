#!/usr/bin/env python3
"""TEST TOOL (CPU): BAM records damaged INSIDE intact BGZF members -- bits flipped, bytes set, 32-bit words overwritten anywhere behind the
header -- through the oracle and through the command on the device stand-in (tools/dev_standin.c); reports every case in which the two
exit codes differ, the command hangs or dies on a signal, and keeps the file.  What it found in round 5 (250 files): one class, a record
whose refID lies outside the header's contigs.
usage: cd DIR_WITH s.fa s.bam dump.tsv (the oracle's MDK_ORACLE_DUMP of the intact file); fuzz_damaged_records.py SEED N"""
import gzip, os, struct, subprocess, sys, zlib
import random
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d=gzip.open('s.bam','rb').read()
l_text=struct.unpack_from('<i',d,4)[0]; o=8+l_text
n_ref=struct.unpack_from('<i',d,o)[0]; o+=4
for _ in range(n_ref):
    ln=struct.unpack_from('<i',d,o)[0]; o+=4+ln+4
hdr_end=o
def bgzf(data):
    out=bytearray()
    for i in range(0,len(data),65280):
        blk=data[i:i+65280]; c=zlib.compressobj(6,zlib.DEFLATED,-15); comp=c.compress(blk)+c.flush()
        bs=len(comp)+25
        out+=b'\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0'+struct.pack('<H',bs)+comp+struct.pack('<II',zlib.crc32(blk),len(blk))
    out+=bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')
    return bytes(out)
env=dict(os.environ, LD_PRELOAD=REPO + '/tools/_build/libmdk_dev_standin.so', MDK_STANDIN_DUMP=os.path.abspath('dump.tsv'))
rnd=random.Random(int(sys.argv[1]) if len(sys.argv)>1 else 1)
bad=0
for it in range(int(sys.argv[2]) if len(sys.argv)>2 else 60):
    b=bytearray(d)
    k=rnd.choice([1,1,2,8])
    for _ in range(k):
        p=rnd.randrange(hdr_end,len(b)); 
        mode=rnd.random()
        if mode<0.5: b[p]^=1<<rnd.randrange(8)
        elif mode<0.8: b[p]=rnd.choice([0,255,128,1])
        else:
            # damage a block_size word: find a record start near p is hard; just write 4 bytes
            b[p:p+4]=struct.pack('<i',rnd.choice([-1,0,3,31,32,2**31-1,1<<20]))
    open('fz.bam','wb').write(bgzf(bytes(b)))
    try:
        o=subprocess.run([REPO + '/oracle/_build/mdk_oracle','extract','s.fa','fz.bam','-o','fo'],capture_output=True,text=True,timeout=60); orc=o.returncode
    except subprocess.TimeoutExpired: orc='HANG'
    try:
        p_=subprocess.run([REPO + '/methyldackel_amd/_build/MethylDackel','extract','s.fa','fz.bam','-@','4','-o','fp'],capture_output=True,text=True,timeout=60,env=env); prc=p_.returncode; perr=(p_.stderr.strip().splitlines() or [''])[-1][:140]
    except subprocess.TimeoutExpired: prc='HANG'; perr=''
    if prc!=orc or prc=='HANG' or (isinstance(prc,int) and (prc<0 or prc in (134,139))):
        bad+=1; print(it,'oracle',orc,(o.stderr.strip().splitlines() or [''])[-1][:100] if orc!='HANG' else '','| product',prc,perr, flush=True)
        os.rename('fz.bam',f'keep_{sys.argv[1] if len(sys.argv)>1 else 1}_{it}.bam')
print('done, differing:',bad)
