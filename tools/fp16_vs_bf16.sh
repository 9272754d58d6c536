#!/bin/bash
# the 64^2 forward launches of an edit in both dtypes, back to back (us per launch): tools/attn_one.py TIME=1
export TIME=1
for DT in bf16 fp16; do
  export DT
  QS=1 BH=5 python3 tools/attn_one.py 200
  QS=1 BH=15 python3 tools/attn_one.py 200
  QS=1 BH=20 python3 tools/attn_one.py 200
  FORM=cfg python3 tools/attn_one.py 200
  FORM=opt python3 tools/attn_one.py 200
  FORM=opt LSUM=0 python3 tools/attn_one.py 200
done
