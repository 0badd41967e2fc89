// acr::vector<T> -- the reference's array handle (src/utils/vector.h:11-44).
//
// In the reference this is a raw {device pointer, size} pair filled by
// cudaMalloc + thrust::copy in its constructor and never freed.  On this side
// of the boundary the host scene stays on the host: the handle keeps the same
// interface (operator[], size, clear, free) over host storage, and the one
// upload happens later, in bulk, through acr_hip_upload_scene.
#ifndef ACR_UTILS_VECTOR_H
#define ACR_UTILS_VECTOR_H

#include <cstddef>
#include <memory>
#include <vector>

namespace acr
{
	template <typename T>
	class vector
	{
	public:
		vector() = default;
		vector(const vector<T> &v) = default;
		// the reference's vector(const thrust::host_vector<T>&) (vector.h:46-55)
		vector(const std::vector<T> &h) : store(std::make_shared<std::vector<T>>(h)) {}

		T &operator[](size_t pos) { return (*store)[pos]; }
		const T &operator[](size_t pos) const { return (*store)[pos]; }
		size_t size() const { return store ? store->size() : 0; }
		const T *data() const { return store ? store->data() : nullptr; }
		void clear() { store.reset(); }
		void free() { store.reset(); }

	private:
		// shared so that copies of the handle alias one array, as copies of the
		// reference's pointer/size pair do
		std::shared_ptr<std::vector<T>> store;
	};
} // namespace acr

#endif
