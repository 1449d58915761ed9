import sys; sys.path.insert(0,'tests')
import selftest as st
r = st.check_step(H=1024, A=16, F_=4096, L=24, S=512, V=250002, std=0.02, bf16_oracle=(True,"flash","flash_split"))
for k,v in r.items():
    if k!="grad_table_top": print(k, v)
