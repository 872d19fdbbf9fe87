# fit() + evaluate() on the Markov-signal synthetic data across model configurations (tools/learn_probe.py); every line should end near
# recall@20 0.85-0.89 (the signal is 90 % deterministic) EXCEPT the 3-layer GRU4Rec, which stays at the 0.07 popularity baseline — as the
# same model does in plain torch (nn.GRU(64, 256, 3, bias=False), same data / loss / Adam: recall@20 0.09 after 10 epochs, loss 1.41):
# a property of the bias-free 3-layer stack at this initialisation, not of the kernels (its gradients match the oracle, tests/test_gpu_gru.py).
run() { echo "== $*"; env "$@" timeout 300 python tools/learn_probe.py 2>&1 | grep -v "^  File" | grep -E "^\{|Error|error|fault" | tail -2 | cut -c1-260; }
run MODEL=SASRec EPOCHS=10 'OVERRIDES={"model":{"layer_num":3,"head_num":1,"hidden_size":256}}'
run MODEL=SASRec EPOCHS=10 'OVERRIDES={"model":{"layer_num":1,"embed_dim":128,"hidden_size":128},"data":{"max_seq_len":20}}'
run MODEL=SASRec EPOCHS=10 'OVERRIDES={"model":{"dropout_rate":0.0},"train":{"batch_size":1000,"hip_graph":false}}'
run MODEL=SASRec EPOCHS=10 'OVERRIDES={"train":{"batch_size":5000},"eval":{"batch_size":64}}'
run MODEL=GRU4Rec EPOCHS=25 'OVERRIDES={"model":{"hidden_size":128,"layer_num":1}}'
run MODEL=GRU4Rec EPOCHS=25 'OVERRIDES={"model":{"layer_num":3},"train":{"batch_size":500}}'
run MODEL=FMLP EPOCHS=25 PREFIX=1 'OVERRIDES={"model":{"layer_num":1}}'
run MODEL=MetaModel SUB=SASRec EPOCHS=14 'OVERRIDES={"train":{"interval":7,"warmup_epoch":3}}'
run MODEL=MetaModel SUB=GRU4Rec EPOCHS=25 WARM=5 
run MODEL=CL4SRec EPOCHS=10
