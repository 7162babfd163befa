/* the code object of k_blockw (asm/blockw.hsaco), linked into libp3hip.so as read-only data */
	.section .rodata
	.global p3_blockw_hsaco
	.global p3_blockw_hsaco_end
	.balign 4096
p3_blockw_hsaco:
	.incbin "asm/blockw.hsaco"
p3_blockw_hsaco_end:
	.byte 0
	.section .note.GNU-stack,"",@progbits
