set pagination off
handle SIGUSR1 nostop noprint
run
echo ---- vector parallel_streams (begin end cap)\n
x/3gx $r12
echo ---- stream pointers\n
x/6gx *(long*)$r12
echo ---- parallel lists vector begin/end at r13+0xe8\n
x/2gx $r13+0xe8
echo ---- launch stream field 0x1a8 and object\n
x/gx $r15+0x1a8
x/16gx *(long*)($r15+0x1a8)
echo ---- parallel stream 0 field and object\n
x/gx *(long*)(*(long*)$r12)+0x1a8
x/16gx *(long*)(*(long*)(*(long*)$r12)+0x1a8)
echo ---- parallel stream 1 field and object\n
x/gx *(long*)(*(long*)$r12+8)+0x1a8
x/16gx *(long*)(*(long*)(*(long*)$r12+8)+0x1a8)
echo ---- disassemble head\n
x/40i $rip-177
echo ---- class of the object at stream+0x1a8 (vtable symbol) and the compared virtual function\n
info symbol *(long*)(*(long*)($r15+0x1a8))
info symbol *(long*)(*(long*)(*(long*)($r15+0x1a8))+0x10)
x/16i *(long*)(*(long*)(*(long*)($r15+0x1a8))+0x10)
