"""Linear-replacement classes of the quantized inference path (reference: quantize/{gptq,bnb}.py)."""
