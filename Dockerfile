# ROCm 7 + PyTorch base; the framework itself is bind-mounted at /distributed
# (see docker-compose.yml) and its gfx950 kernels are built at image build time.
FROM rocm/pytorch:rocm7.0_ubuntu22.04_py3.10_pytorch_release_2.8.0

RUN pip install --no-cache-dir fastapi uvicorn msgpack cloudpickle prometheus_client \
    safetensors transformers jinja2 httpx

WORKDIR /distributed
COPY . /distributed
RUN python3 -m vllm_distributed_amd.build
