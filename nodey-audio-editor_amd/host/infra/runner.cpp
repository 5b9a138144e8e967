#include "runner.hpp"

namespace infra
{
	std::map<std::string, Processor::Info> Processor::processor_map;  // reference: src/infra/processor.cpp:5

	void Runner::add_node(Id_t id, std::shared_ptr<Processor> processor)
	{
		auto resource = std::make_shared<Processor_resource>();
		resource->processor = std::move(processor);
		for (const auto& pin : resource->processor->get_pin_attributes())
			if (!pin.is_input) resource->output_payloads.emplace(pin.identifier, std::set<std::shared_ptr<Processor::Product>>{});
		processor_resources[id] = std::move(resource);
	}

	void Runner::add_link(const Link& link) { links.push_back(link); }

	void Runner::request_stop()
	{
		for (auto& [_, resource] : processor_resources) resource->stop_source = true;
	}

	bool Runner::run()
	{
		for (const auto& link : links)
		{
			auto& from = processor_resources.at(link.from);
			auto& to = processor_resources.at(link.to);
			std::shared_ptr<Processor::Product> product;
			for (const auto& pin : from->processor->get_pin_attributes())
				if (!pin.is_input && pin.identifier == link.from_pin) product = pin.generate_func();
			if (!product) THROW_LOGIC_ERROR("no output pin '%s' on node %d", link.from_pin.c_str(), link.from);
			from->output_payloads[link.from_pin].emplace(product);
			if (!to->input_payloads.emplace(link.to_pin, product).second)
				THROW_LOGIC_ERROR("input pin '%s' of node %d has two links", link.to_pin.c_str(), link.to);  // graph.cpp:180-282
		}
		nae_fiber::Scheduler scheduler;
		for (auto& [idx, resource] : processor_resources)
		{
			scheduler.spawn(
				[this, ptr = resource.get(), idx = idx]
				{
					const auto find_node_data = node_data.find(idx);
					std::any fallback;
					try
					{
						ptr->state = State::Running;
						ptr->processor->process_payload(
							ptr->input_payloads,
							ptr->output_payloads,
							ptr->stop_source,
							find_node_data == node_data.end() ? fallback : *(find_node_data->second)
						);
						ptr->state = State::Finished;
					}
					catch (const Processor::Runtime_error& e)
					{
						ptr->exception = e;
						ptr->error_text = e.what();
						ptr->state = State::Error;
					}
					catch (const std::exception& e)
					{
						ptr->exception = std::runtime_error(e.what());
						ptr->error_text = e.what();
						ptr->state = State::Error;
					}
					catch (...)
					{
						ptr->exception = std::exception();
						ptr->error_text = "unknown exception";
						ptr->state = State::Error;
					}
					// the GUI tears the whole Runner down on the first error (frontend/app.cpp:1929-1935)
					if (ptr->state == State::Error) request_stop();
				}
			);
		}
		scheduler.run();
		switches = scheduler.switches();
		for (auto& [_, resource] : processor_resources)
			if (resource->state == State::Error) return false;
		return true;
	}
}
