"""ViT on the MI355X HIP kernels (placeholder until the transformer kernels land)."""
__all__ = []
