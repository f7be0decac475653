"""oracle/ — TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch ops on the host, fp32 or bf16-autocast) of the reference's image/
hot path, written functionally over a state-dict so it shares no code with the product (reed_amd/).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it — as the checker /
reported baseline, never as the thing measured or shipped.

Pinning: the reference ships no tests or golden vectors for this path, so the oracle
is pinned against outputs of the reference itself, generated in the authoring container by
tools/gen_golden.py (imports /root/reference/image with a stand-in for the un-vendored `timm`
classes PatchEmbed/Attention/Mlp) and committed as small fixtures under tests/golden/.
tests/test_oracle_golden.py checks every oracle function against them.
"""
